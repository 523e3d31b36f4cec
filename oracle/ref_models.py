"""ORACLE (test infrastructure only) — CPU fp32 restatement of the reference's
joint segmentation + disparity graphs in plain torch.nn.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package never does.  Module/attribute names reproduce
the reference's state_dict keys so that weights captured from the reference load
unchanged; the arithmetic follows the cited reference lines.  The restatement is
pinned against the reference itself by tests/golden/*.npz (oracle/make_golden.py
imports /root/reference in the build container and records inputs/outputs).

Exception: the correlation operator is third-party and absent (see
oracle/corr_ref.c) — "parity unpinned" at that boundary.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- operators
class SpatialCorrelationSampler(nn.Module):
    """kernel_size=1, stride=1, padding=0 sampler (models/dsnet_t2.py:1078-1087); see oracle/corr_ref.c."""

    def __init__(self, kernel_size=1, patch_size=1, stride=1, padding=0, dilation=1, dilation_patch=1):
        super().__init__()
        assert kernel_size == 1 and stride == 1 and padding == 0 and dilation == 1, "only the configuration the reference uses"
        self.patch = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.dil = dilation_patch

    def forward(self, a, b):
        PH, PW = self.patch
        B, C, H, W = a.shape
        rh, rw = PH // 2 * self.dil, PW // 2 * self.dil
        bp = F.pad(b, (rw, rw, rh, rh))
        rows = []
        for ph in range(PH):
            cols = []
            for pw in range(PW):
                sh = bp[:, :, ph * self.dil: ph * self.dil + H, pw * self.dil: pw * self.dil + W]
                cols.append((a * sh).sum(1))
            rows.append(torch.stack(cols, 1))
        return torch.stack(rows, 1)  # (B, PH, PW, H, W)


def _tf_same_pad(size, stride, k, dil):
    """models/torch_model.py:276-281."""
    out = math.ceil(size / float(stride))
    total = max((out - 1) * stride - size + dil * (k - 1) + 1, 0)
    lo = int(total // 2)
    return lo, int(total - lo)


def _he_init(mods):
    """models/dsnet_t2.py:37-43."""
    for m in mods:
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            m.weight.data.normal_(0, math.sqrt(2.0 / (m.kernel_size[0] * m.kernel_size[1] * m.out_channels)))
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()


class conv2dSame(nn.Module):
    """models/torch_model.py:236-281."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding='valid', dilation=1, bias=False):
        super().__init__()
        self.padding = padding
        self.c2d = nn.Conv2d(in_channel, out_channel, kernel_size, stride=stride, dilation=dilation, bias=bias)
        _he_init([self.c2d])

    def forward(self, x):
        if self.padding == 'same':
            s, k, d = self.c2d.stride[0], self.c2d.kernel_size[0], self.c2d.dilation[0]
            t, b = _tf_same_pad(x.shape[2], s, k, d)
            l, r = _tf_same_pad(x.shape[3], s, k, d)
            x = F.pad(x, (l, r, t, b))
        return self.c2d(x)


class ConvTranspose2dSame(nn.Module):
    """models/torch_model.py:284-349: full transposed conv, then centre crop to H*stride."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding='valid', dilation=1, bias=False,
                 init_he=True):
        super().__init__()
        self.padding = padding
        self.ct2d = nn.ConvTranspose2d(in_channel, out_channel, kernel_size, stride=stride, dilation=dilation, bias=bias)

    def forward(self, x):
        if self.padding != 'same':
            return self.ct2d(x)
        th, tw = x.shape[2] * self.ct2d.stride[0], x.shape[3] * self.ct2d.stride[1]
        x = self.ct2d(x)
        h, w = x.shape[2] // 2, x.shape[3] // 2
        oh = th // 2 if h - th // 2 >= 0 else h
        ow = tw // 2 if w - tw // 2 >= 0 else w
        return x[:, :, h - oh: h + th - oh, w - ow: w + tw - ow]


class convbn(nn.Module):
    """models/dsnet_t2.py:16-46."""

    def __init__(self, in_channel, out_channel, kernel_size, stride, pad, dilation, batchnorm=True):
        super().__init__()
        seq = [conv2dSame(in_channel, out_channel, kernel_size, stride, pad, dilation, bias=not batchnorm)]
        if batchnorm:
            seq.append(nn.BatchNorm2d(out_channel))
        self.layers = nn.Sequential(*seq)
        _he_init(self.modules())

    def forward(self, x):
        return self.layers(x)


class deconvbn(nn.Module):
    """models/dsnet_t2.py:48-77."""

    def __init__(self, in_channel, out_channel, kernel_size, stride, pad, dilation, batchnorm=True):
        super().__init__()
        seq = [ConvTranspose2dSame(in_channel, out_channel, kernel_size, stride, pad, dilation, bias=not batchnorm)]
        if batchnorm:
            seq.append(nn.BatchNorm2d(out_channel))
        self.layers = nn.Sequential(*seq)
        _he_init(self.modules())

    def forward(self, x):
        return self.layers(x)


def _act(block, p=0.0):
    return nn.Sequential(block, nn.ReLU(inplace=True), nn.Dropout(p=p))


class Conv2DownUp(nn.Module):
    """models/dsnet_t2.py:80-117."""

    def __init__(self, in_channels, out_channels=3, kernel_size=3, lastLayer=True, dropout=0):
        super().__init__()
        self.lastLayer = lastLayer
        o, k = out_channels, kernel_size
        self.c1 = _act(convbn(in_channels, o, k, 1, 'same', 1), dropout)
        self.c2 = _act(convbn(o, o, k, 1, 'same', 1), dropout)
        self.c3 = _act(convbn(o, o, k, 1, 'same', 1), dropout)
        self.d3 = _act(deconvbn(o, o, k, 1, 'same', 1), dropout)
        self.d4 = _act(deconvbn(o, o, k, 1, 'same', 1), dropout)
        self.d5 = _act(deconvbn(o, o, k, 1, 'same', 1), dropout)

    def forward(self, x):
        x1 = self.c1(x)
        x2 = self.c2(x1)
        x = x2 + self.d3(self.c3(x2))
        x = x1 + self.d4(x)
        return self.d5(x) if self.lastLayer else x


# --------------------------------------------------------------------------- DenseNet (models/densenet.py)
class _DenseLayer(nn.Module):
    """models/densenet.py:25-93 (drop_rate 0, not memory-efficient)."""

    def __init__(self, cin, growth, bn_size):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(cin)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(cin, bn_size * growth, 1, bias=False)
        self.norm2 = nn.BatchNorm2d(bn_size * growth)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(bn_size * growth, growth, 3, padding=1, bias=False)

    def forward(self, feats):
        x = torch.cat(feats, 1)
        x = self.conv1(self.relu1(self.norm1(x)))
        return self.conv2(self.relu2(self.norm2(x)))


class _DenseBlock(nn.ModuleDict):
    """models/densenet.py:96-116."""

    def __init__(self, n, cin, bn_size, growth):
        super().__init__()
        for i in range(n):
            self['denselayer%d' % (i + 1)] = _DenseLayer(cin + i * growth, growth, bn_size)

    def forward(self, x):
        feats = [x]
        for layer in self.values():
            feats.append(layer(feats))
        return torch.cat(feats, 1)


class _Transition(nn.Sequential):
    """models/densenet.py:119-128 (pool moved out of the transition)."""

    def __init__(self, cin, cout):
        super().__init__(OrderedDict(norm=nn.BatchNorm2d(cin), relu=nn.ReLU(inplace=True),
                                     conv=nn.Conv2d(cin, cout, 1, bias=False)))


class DenseNet(nn.Module):
    """models/densenet.py:131-245: five taps, tap 0 is the raw conv0 output."""

    def __init__(self, growth_rate=32, block_config=(6, 12, 24, 16), num_init_features=64, bn_size=4, num_classes=1000):
        super().__init__()
        self.conv0 = nn.Conv2d(3, num_init_features, 7, stride=2, padding=3, bias=False)
        self.features = nn.Sequential(OrderedDict(norm0=nn.BatchNorm2d(num_init_features), relu0=nn.ReLU(inplace=True),
                                                  pool0=nn.MaxPool2d(3, stride=2, padding=1)))
        blocks, c = [], num_init_features
        for i, n in enumerate(block_config):
            blocks.append(_DenseBlock(n, c, bn_size, growth_rate))
            c += n * growth_rate
            if i != len(block_config) - 1:
                blocks.append(_Transition(c, c // 2))
                c //= 2
        self.denseblock = nn.ModuleList(blocks)
        self.norm5 = nn.BatchNorm2d(c)
        self.classifier = nn.Linear(c, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        c0 = self.conv0(x)
        taps = [c0]
        f = self.features(c0)
        for i, blk in enumerate(self.denseblock):
            f = blk(f)
            if i % 2:
                taps.append(f)
                f = F.avg_pool2d(f, 2, 2)
        taps.append(F.relu(self.norm5(f)))
        return taps


def densenet121(pretrained=False):
    assert not pretrained, "no network access: construct with pretrained=False"
    return DenseNet(32, (6, 12, 24, 16), 64)


# --------------------------------------------------------------------------- pyramids
def _pool_branch(p, cin):
    return nn.Sequential(nn.AvgPool2d(p, p), convbn(cin, 32, 3, 1, 'same', 1), nn.ReLU(inplace=True))


def _pyr(branches, x):
    return torch.cat([x] + [F.interpolate(b(x), x.shape[2:], mode='bilinear') for b in branches], 1)


class piramidNet(nn.Module):
    """models/dsnet_t2.py:324-390 (pyramid of dsnet): taps + b2 (tap 2) + b0 (tap 0)."""

    def __init__(self, pretrained=False):
        super().__init__()
        self.resnet_features = densenet121(pretrained)
        pv = [128, 64, 32, 16, 8]
        for j in range(5):
            setattr(self, 'branch0_%d' % j, _pool_branch(pv[j], 64))
        for j in range(3):
            setattr(self, 'branch1_%d' % j, _pool_branch(pv[j + 2], 256))

    def forward(self, x):
        o = self.resnet_features(x)
        b0 = _pyr([getattr(self, 'branch0_%d' % j) for j in range(5)], o[0])
        b2 = _pyr([getattr(self, 'branch1_%d' % j) for j in range(3)], o[2])
        return o[0], o[1], o[2], o[3], o[4], b2, b0


class piramidNet2(nn.Module):
    """models/dsnet_t2.py:1893-2083, densenet backbone."""

    def __init__(self, pretrained=False, backbone='densenet'):
        super().__init__()
        assert backbone == 'densenet'
        self.backbone = backbone
        self.resnet_features = densenet121(pretrained)
        pv, cin = [128, 64, 32, 16, 8], [64, 128, 256]
        for j in range(5):
            setattr(self, 'branch0_%d' % j, _pool_branch(pv[j], cin[0]))
        for j in range(4):
            setattr(self, 'branch1_%d' % j, _pool_branch(pv[j + 1], cin[1]))
        for j in range(3):
            setattr(self, 'branch2_%d' % j, _pool_branch(pv[j + 2], cin[2]))

    def forward(self, x):
        o = self.resnet_features(x)
        b0 = _pyr([getattr(self, 'branch0_%d' % j) for j in range(5)], o[0])
        b1 = _pyr([getattr(self, 'branch1_%d' % j) for j in range(4)], o[1])
        b2 = _pyr([getattr(self, 'branch2_%d' % j) for j in range(3)], o[2])
        return o[0], o[1], o[2], o[3], o[4], b2, b1, b0


# --------------------------------------------------------------------------- ASPP (models/aspp.py)
class _ASPPModule(nn.Module):
    def __init__(self, cin, cout, k, padding, dilation):
        super().__init__()
        self.atrous_conv = nn.Conv2d(cin, cout, k, stride=1, padding=padding, dilation=dilation, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU()
        nn.init.kaiming_normal_(self.atrous_conv.weight)

    def forward(self, x):
        return self.relu(self.bn(self.atrous_conv(x)))


class ASPP(nn.Module):
    """models/aspp.py:34-108."""
    _INPLANES = {'drn': 512, 'mobilenet': 320, 'densenet_a1': 128, 'densenet_a3': 512, 'mobilenet_a1': 24,
                 'mobilenet_a3': 112, 'resnet50_a1': 256, 'resnet50_a3': 1024, 'resnet50_a4': 2048}
    _DIL = {32: [1, 2, 6, 12], 16: [1, 6, 12, 18], 8: [1, 12, 24, 36]}

    def __init__(self, backbone, output_stride, BatchNorm=nn.BatchNorm2d):
        super().__init__()
        cin = self._INPLANES.get(backbone, 2048)
        d = self._DIL[output_stride]
        self.aspp1 = _ASPPModule(cin, 256, 1, 0, d[0])
        self.aspp2 = _ASPPModule(cin, 256, 3, d[1], d[1])
        self.aspp3 = _ASPPModule(cin, 256, 3, d[2], d[2])
        self.aspp4 = _ASPPModule(cin, 256, 3, d[3], d[3])
        self.global_avg_pool = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), nn.Conv2d(cin, 256, 1, stride=1, bias=False),
                                             nn.BatchNorm2d(256), nn.ReLU())
        self.conv1 = nn.Conv2d(1280, 256, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(256)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(0.5)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)

    def forward(self, x):
        x5 = F.interpolate(self.global_avg_pool(x), size=x.shape[2:], mode='bilinear', align_corners=True)
        x = torch.cat((self.aspp1(x), self.aspp2(x), self.aspp3(x), self.aspp4(x), x5), 1)
        return self.dropout(self.relu(self.bn1(self.conv1(x))))


def build_aspp(backbone, output_stride, BatchNorm=nn.BatchNorm2d):
    return ASPP(backbone, output_stride, BatchNorm)


# --------------------------------------------------------------------------- heads and full nets
def _c1x1(cin, cout):
    return nn.Sequential(conv2dSame(cin, cout, 1, padding='same'), nn.ReLU(inplace=True))


def _img_conv(cin):
    return nn.Sequential(convbn(cin, 1, 5, 1, 'same', 2), nn.ReLU(inplace=True))


class segNet(nn.Module):
    """models/dsnet_t2.py:915-938."""

    def __init__(self, in_channels, feature_channel, labels=8, pretrained=False, dropout=0):
        super().__init__()
        self.conv1d_1 = _c1x1(in_channels, 64)
        self.Conv2DownUp1 = Conv2DownUp(64, 32, 3, dropout=dropout)
        self.conv1d_2 = _c1x1(32 + feature_channel, 32)
        self.Conv2DownUp2 = nn.Sequential(Conv2DownUp(32, 32, 3, lastLayer=False, dropout=dropout),
                                          ConvTranspose2dSame(32, labels, 3, 1, padding='same', init_he=False))

    def forward(self, x, input_a, input_b, xleft):
        x = F.interpolate(x, scale_factor=2, mode='nearest')
        x = self.Conv2DownUp1(self.conv1d_1(x))
        x1 = F.interpolate(x, scale_factor=2, mode='nearest')
        s = F.interpolate(x, size=xleft.shape[2:], mode='nearest')
        s = self.Conv2DownUp2(self.conv1d_2(torch.cat((s, xleft), 1)))
        return x, x1, F.interpolate(s, size=input_a.shape[2:], mode='nearest')


class CFG:
    """The fields minidsnetExt reads from the argparse namespace (models/dsnet_t2.py:944-953)."""

    def __init__(self, dropout=0.0, multaskloss=0, aspp=0, use_att=1, hanet=0, convDeconvOut=0, abilation=''):
        self.dropout, self.multaskloss, self.aspp, self.use_att = dropout, multaskloss, aspp, use_att
        self.hanet, self.convDeconvOut, self.abilation = hanet, convDeconvOut, abilation


# --------------------------------------------------------------------------- HANet (models_hanet/*)
def get_sinusoid_encoding_table(n_position, d_hid):
    """models_hanet/PosEmbedding.py:7-28 (the reference's last assignment wins: cycle = 10 if d_hid > 50 else 100)."""
    import numpy as np
    cycle = 10 if d_hid > 50 else 100
    table = np.array([[pos / np.power(cycle, 2 * (j // 2) / d_hid) for j in range(d_hid)] for pos in range(n_position)])
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    return torch.FloatTensor(table)


class PosEncoding1D(nn.Module):
    """models_hanet/PosEmbedding.py:49-85 without the hard-coded .cuda() calls (pos_noise = 0)."""

    def __init__(self, pos_rfactor, dim, pos_noise=0.0):
        super().__init__()
        assert pos_noise == 0.0
        self.pos_layer = nn.Embedding.from_pretrained(get_sinusoid_encoding_table((128 // pos_rfactor) + 1, dim) + 1, freeze=True)
        self.pos_rfactor = pos_rfactor

    def forward(self, x, pos):
        pos_h, _ = pos
        pos_h = pos_h // self.pos_rfactor
        pos_h = pos_h.index_select(2, torch.tensor([0])).unsqueeze(1).squeeze(3)
        pos_h = F.interpolate(pos_h.float(), size=x.shape[2], mode='nearest').long()
        return x + self.pos_layer(pos_h).transpose(1, 3).squeeze(3)


class HANet_Conv(nn.Module):
    """models_hanet/HANet.py:9-128 (pooling 'max' | 'mean', sinusoid encoding, the attention_loss return form)."""

    def __init__(self, in_channel, out_channel, kernel_size=3, r_factor=64, layer=3, pos_injection=2, is_encoding=1,
                 pos_rfactor=8, pooling='mean', dropout_prob=0.0, pos_noise=0.0):
        super().__init__()
        import math
        assert is_encoding == 1
        self.pooling, self.pos_injection, self.layer, self.dropout_prob = pooling, pos_injection, layer, dropout_prob
        self.sigmoid = nn.Sigmoid()
        mid_1 = math.ceil(in_channel / r_factor) if r_factor > 0 else in_channel * (-r_factor)
        if dropout_prob > 0:
            self.dropout = nn.Dropout2d(dropout_prob)
        self.attention_first = nn.Sequential(nn.Conv1d(in_channel, mid_1, 1, bias=False), nn.BatchNorm1d(mid_1), nn.ReLU(inplace=True))
        if layer == 2:
            self.attention_second = nn.Sequential(nn.Conv1d(mid_1, out_channel, kernel_size, padding=kernel_size // 2, bias=True))
        else:
            mid_2 = mid_1 * 2
            self.attention_second = nn.Sequential(nn.Conv1d(mid_1, mid_2, 3, padding=1, bias=True), nn.BatchNorm1d(mid_2),
                                                  nn.ReLU(inplace=True))
            self.attention_third = nn.Sequential(nn.Conv1d(mid_2, out_channel, kernel_size, padding=kernel_size // 2, bias=True))
        rows = 128 // pos_rfactor
        self.rowpool = nn.AdaptiveAvgPool2d((rows, 1)) if pooling == 'mean' else nn.AdaptiveMaxPool2d((rows, 1))
        if pos_rfactor > 0:
            if pos_injection == 1:
                self.pos_emb1d_1st = PosEncoding1D(pos_rfactor, dim=in_channel, pos_noise=pos_noise)
            else:
                self.pos_emb1d_2nd = PosEncoding1D(pos_rfactor, dim=mid_1, pos_noise=pos_noise)

    def forward(self, x, out, pos=None, return_attention=False, return_posmap=False, attention_loss=False):
        H = out.size(2)
        x1d = self.rowpool(x).squeeze(3)
        if pos is not None and self.pos_injection == 1:
            x1d = self.pos_emb1d_1st(x1d, pos)
        if self.dropout_prob > 0:
            x1d = self.dropout(x1d)
        x1d = self.attention_first(x1d)
        if pos is not None and self.pos_injection == 2:
            x1d = self.pos_emb1d_2nd(x1d, pos)
        x1d = self.attention_second(x1d)
        if self.layer == 3:
            x1d = self.attention_third(x1d)
        last_attention = x1d
        x1d = self.sigmoid(x1d)
        x1d = F.interpolate(x1d, size=H, mode='linear')
        out = torch.mul(out, x1d.unsqueeze(3))
        if return_attention:
            return out, x1d
        if attention_loss:
            return out, last_attention
        return out


class minidsnetExt(nn.Module):
    """models/dsnet_t2.py:941-1299 — densenet backbone, no multitask loss; HANet head as upstream (built when
    CFG.hanet, applied only on the non-aspp-2 branch: dsnet_t2.py:1135-1150,1287-1289)."""

    def __init__(self, CFG, labels=8, pretrained=False, patch_type='', include_edges=False, backbone='densenet'):
        super().__init__()
        assert backbone == 'densenet' and not CFG.multaskloss
        self.include_edges = include_edges
        self.hanet = CFG.hanet
        dropout = CFG.dropout
        self.aspp_mod, self.use_att, self.convDeconvOut, self.abilation = CFG.aspp, CFG.use_att, CFG.convDeconvOut, CFG.abilation
        self.patch_type, self.backbone = patch_type, backbone
        feature_channel, inplane_seg2 = 1, 512
        if self.aspp_mod == 1:
            self.aspp, inplane_seg2 = build_aspp('densenet_a1', 32), 256
        elif self.aspp_mod == 2:
            self.aspp, inplane_seg2, feature_channel = build_aspp('densenet_a3', 32), 273, 64
        self.resnet_features = piramidNet2(pretrained, backbone)
        for j in range(4):      # aux_img_channel = 4 with the edge map (models/dsnet_t2.py:1061-1069)
            setattr(self, 'conv2d_ba%d' % j, _img_conv(4 if include_edges else 3))
        patch = (1, 17) if patch_type == '1dcorr' else (17, 17)
        self.correlation_sampler = SpatialCorrelationSampler(1, patch, 1, 0, dilation_patch=1)
        self.s2_corr_sampler = SpatialCorrelationSampler(1, patch, 1, 0, dilation_patch=1)
        self.corrConv2d = _c1x1(patch[0] * patch[1], 128)
        self.Conv2DownUp3 = Conv2DownUp(352 if 'no_dec1' in self.abilation else 32, 128, 3, dropout=dropout)
        self.Conv2DownUp4 = Conv2DownUp(256, 64, 3, dropout=dropout)
        self.segNet = segNet(2048, 1, labels, dropout=dropout)
        self.conv1d_2 = _c1x1(65, 64)
        self.Conv2DownUp5 = Conv2DownUp(64, 64, 5, lastLayer=False, dropout=dropout)
        self.dispoutConv = ConvTranspose2dSame(64, 1, 5, padding='same', init_he=False)
        self.conv1d_3 = _c1x1(96, 64)
        self.conv1d_4 = _c1x1(inplane_seg2, 128)
        self.Conv2DownUp6 = Conv2DownUp(128, 64, 3, dropout=dropout)
        self.Conv2DownUp7 = Conv2DownUp(128, 64, 3, dropout=dropout)
        self.Conv2DownUp8 = Conv2DownUp(32, 64, 3, dropout=dropout)
        self.Conv2DownUp9 = Conv2DownUp(128, 64, 3, dropout=dropout)
        self.conv1d_at_d = nn.Sequential(conv2dSame(64, 1, 1, padding='same'), nn.Sigmoid(), nn.Dropout(p=dropout))
        self.conv1d_at_s = nn.Sequential(conv2dSame(64, 1, 1, padding='same'), nn.Sigmoid(), nn.Dropout(p=dropout))
        c10 = 64 if 'no_dec3' in self.abilation else (128 if self.use_att else 192)
        self.Conv2DownUp10 = Conv2DownUp(c10, 64, 3, dropout=dropout)
        self.conv1d_5 = _c1x1(64 + feature_channel, 32)
        if self.convDeconvOut:
            self.Conv2DownUp11 = nn.Sequential(Conv2DownUp(32, 32, 3, lastLayer=False))
            self.convOutput2 = conv2dSame(32, labels, 3, 1, padding='same')
            if self.convDeconvOut == 2:
                self.convOutput = ConvTranspose2dSame(32, labels, 3, 1, padding='same', init_he=False)
        else:
            self.Conv2DownUp11 = nn.Sequential(Conv2DownUp(32, 32, 3, lastLayer=False, dropout=dropout),
                                               ConvTranspose2dSame(32, labels, 3, 1, padding='same', init_he=False))
        if self.hanet:
            self.hanet_last = HANet_Conv(64, labels, pooling='max', pos_rfactor=2, dropout_prob=0.1)
            for m in self.hanet_last.modules():
                if isinstance(m, nn.Conv1d):
                    nn.init.kaiming_normal_(m.weight, nonlinearity='relu')
                    if m.bias is not None:
                        m.bias.data.zero_()
                elif isinstance(m, nn.BatchNorm1d):
                    m.weight.data.fill_(1)
                    m.bias.data.zero_()

    def forward(self, input_a, input_b, pos=None, disp_gt=None, seg_gt=None):
        left, right = (input_a[:, :3], input_b[:, :3]) if self.include_edges else (input_a, input_b)   # models/dsnet_t2.py:1153-1158
        a = self.resnet_features(left)  # a_0..a_4, B2, B1, B0
        b = self.resnet_features(right)
        xl3, xl2 = self.conv2d_ba3(input_a), self.conv2d_ba1(input_a)
        xl1, xl0 = self.conv2d_ba2(input_a), self.conv2d_ba0(input_a)
        x, x1, seg1 = self.segNet(torch.cat([a[4], b[4]], 1), input_a, input_b, xl0)

        y = self.correlation_sampler(a[5], b[5])
        if self.patch_type == '1dcorr':
            y = torch.squeeze(y, 1)
        else:
            n, ph, pw, h, w = y.shape
            y = y.reshape(n, ph * pw, h, w) / a[5].size(1)
        y = self.corrConv2d(y)
        y1 = self.Conv2DownUp3(a[5] if 'no_dec1' in self.abilation else x1)
        y1 = F.interpolate(y1, size=y.shape[2:], mode='bilinear')
        y = self.Conv2DownUp4(torch.cat((y1, y), 1))
        y2 = F.interpolate(y, scale_factor=8)
        xl2 = F.interpolate(xl2, size=y2.shape[2:], mode='bilinear')
        d = self.Conv2DownUp5(self.conv1d_2(torch.cat((y2, xl2), 1)))
        disp = F.interpolate(self.dispoutConv(d), size=input_a.shape[2:], mode='bilinear')

        if self.aspp_mod == 1:
            s2 = self.aspp(a[1])
        elif self.aspp_mod == 2:
            s21, s22 = self.aspp(a[3]), self.aspp(b[3])
            s2 = torch.cat((torch.squeeze(self.s2_corr_sampler(s21, s22), 1), s21), 1)
        else:
            s2 = torch.cat((a[6], b[6]), 1)
        s2 = self.Conv2DownUp6(self.conv1d_4(s2))
        y3 = F.interpolate(y, size=s2.shape[2:])
        if 'no_dec3' not in self.abilation:
            x3 = F.interpolate(self.Conv2DownUp8(x1), size=s2.shape[2:])
            if self.use_att:
                s2_d = self.Conv2DownUp7(torch.cat((s2, y3), 1))
                at_d = self.conv1d_at_d(s2_d)
                s2_s = self.Conv2DownUp9(torch.cat((s2, x3), 1))
                at_s = self.conv1d_at_s(s2_s)
                s2 = torch.cat((s2_d * at_s, s2_s * at_d), 1)
            else:
                s2 = torch.cat((s2, x3, y3), 1)
        s2 = self.Conv2DownUp10(s2)
        if self.aspp_mod == 2:
            s2 = torch.cat((F.interpolate(s2, size=a[0].shape[2:]), a[0]), 1)
            seg2 = self.Conv2DownUp11(self.conv1d_5(s2))
            seg2 = F.interpolate(seg2, size=input_a.shape[2:], mode='nearest')
        else:
            s2 = torch.cat([F.interpolate(s2, size=xl1.shape[2:]), xl1], 1)
            seg2 = self.Conv2DownUp11(self.conv1d_5(s2))
            if self.convDeconvOut:
                s = self.convOutput2(seg2)
                seg2 = (self.convOutput(seg2) + s) if self.convDeconvOut == 2 else s
            if self.hanet:
                seg2, _ = self.hanet_last(a[0], seg2, pos, attention_loss=True)
        return seg1, disp, seg2, disp


class minidsnet(nn.Module):
    """models/dsnet_t2.py:825-913 (`sdnet_mini`): piramidNet pyramid, coarse segmentation head, correlation disparity head."""

    def __init__(self, CFG, labels=8, pretrained=False, patch_type='', include_edges=False):
        super().__init__()
        self.patch_type, self.include_edges = patch_type, include_edges
        self.resnet_features = piramidNet(pretrained=pretrained)
        for j in range(4):
            setattr(self, 'conv2d_ba%d' % j, _img_conv(4 if include_edges else 3))
        patch = (1, 17) if patch_type == '1dcorr' else (17, 17)
        self.correlation_sampler = SpatialCorrelationSampler(1, patch, 1, 0, dilation_patch=1)
        self.corrConv2d = _c1x1(patch[0] * patch[1], 128)
        self.Conv2DownUp3 = Conv2DownUp(32, 128, 3)
        self.Conv2DownUp4 = Conv2DownUp(256, 64, 3)
        self.segNet = segNet(2048, 1, labels)
        self.conv1d_2 = _c1x1(65, 64)
        self.Conv2DownUp5 = Conv2DownUp(64, 64, 5, lastLayer=False)
        self.dispoutConv = ConvTranspose2dSame(64, 1, 5, padding='same', init_he=False)
        self.conv1d_3 = _c1x1(96, 64)

    def forward(self, input_a, input_b):
        left, right = (input_a[:, :3], input_b[:, :3]) if self.include_edges else (input_a, input_b)
        a = self.resnet_features(left)      # a_0..a_4, pyramid over tap 2, pyramid over tap 0
        b = self.resnet_features(right)
        self.conv2d_ba3(input_a)            # computed and unused upstream (:868,870): their BatchNorm running statistics move
        xl2 = self.conv2d_ba1(input_a)
        self.conv2d_ba2(input_a)
        xl0 = self.conv2d_ba0(input_a)
        x, x1, seg1 = self.segNet(torch.cat([a[4], b[4]], 1), input_a, input_b, xl0)
        y = self.correlation_sampler(a[5], b[5])
        if self.patch_type == '1dcorr':
            y = torch.squeeze(y, 1)
        else:
            n, ph, pw, h, w = y.shape
            y = y.reshape(n, ph * pw, h, w) / a[5].size(1)
        y = self.corrConv2d(y)
        y1 = F.interpolate(self.Conv2DownUp3(x1), size=y.shape[2:], mode='bilinear')
        y = self.Conv2DownUp4(torch.cat((y1, y), 1))
        y2 = F.interpolate(y, scale_factor=8)
        xl2 = F.interpolate(xl2, size=y2.shape[2:], mode='bilinear')
        d = self.dispoutConv(self.Conv2DownUp5(self.conv1d_2(torch.cat((y2, xl2), 1))))
        disp = F.interpolate(d, size=left.shape[2:], mode='bilinear')
        return seg1, disp, seg1, disp


# --------------------------------------------------------------------------- PSMNet (models_psmnet/*)
def _cb2(cin, cout, k, s, pad, dil):
    """models_psmnet/submodule.py:10-13."""
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride=s, padding=dil if dil > 1 else pad, dilation=dil, bias=False),
                         nn.BatchNorm2d(cout))


def _cb3(cin, cout, k, s, pad):
    """models_psmnet/submodule.py:16-19."""
    return nn.Sequential(nn.Conv3d(cin, cout, k, padding=pad, stride=s, bias=False), nn.BatchNorm3d(cout))


class BasicBlock(nn.Module):
    """models_psmnet/submodule.py:21-46."""
    expansion = 1

    def __init__(self, inplanes, planes, stride, downsample, pad, dilation):
        super().__init__()
        self.conv1 = nn.Sequential(_cb2(inplanes, planes, 3, stride, pad, dilation), nn.ReLU(inplace=True))
        self.conv2 = _cb2(planes, planes, 3, 1, pad, dilation)
        self.downsample, self.stride = downsample, stride

    def forward(self, x):
        out = self.conv2(self.conv1(x))
        return out + (self.downsample(x) if self.downsample is not None else x)


class feature_extraction(nn.Module):
    """models_psmnet/submodule.py:66-141."""

    def __init__(self):
        super().__init__()
        self.inplanes = 32
        self.firstconv = nn.Sequential(_cb2(3, 32, 3, 2, 1, 1), nn.ReLU(inplace=True), _cb2(32, 32, 3, 1, 1, 1), nn.ReLU(inplace=True),
                                       _cb2(32, 32, 3, 1, 1, 1), nn.ReLU(inplace=True))
        self.layer1 = self._make(32, 3, 1, 1, 1)
        self.layer2 = self._make(64, 16, 2, 1, 1)
        self.layer3 = self._make(128, 3, 1, 1, 1)
        self.layer4 = self._make(128, 3, 1, 1, 2)
        for j, p in enumerate((64, 32, 16, 8)):
            setattr(self, 'branch%d' % (j + 1), nn.Sequential(nn.AvgPool2d((p, p), stride=(p, p)), _cb2(128, 32, 1, 1, 0, 1), nn.ReLU(inplace=True)))
        self.lastconv = nn.Sequential(_cb2(320, 128, 3, 1, 1, 1), nn.ReLU(inplace=True), nn.Conv2d(128, 32, 1, bias=False))

    def _make(self, planes, blocks, stride, pad, dilation):
        ds = None
        if stride != 1 or self.inplanes != planes:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride=stride, bias=False), nn.BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, ds, pad, dilation)]
        self.inplanes = planes
        layers += [BasicBlock(planes, planes, 1, None, pad, dilation) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        raw = self.layer2(self.layer1(self.firstconv(x)))
        skip = self.layer4(self.layer3(raw))
        br = [F.interpolate(getattr(self, 'branch%d' % j)(skip), skip.shape[2:], mode='bilinear') for j in (1, 2, 3, 4)]
        return self.lastconv(torch.cat((raw, skip, br[3], br[2], br[1], br[0]), 1))


class hourglass(nn.Module):
    """models_psmnet/stackhourglass.py:10-50."""

    def __init__(self, c):
        super().__init__()
        self.conv1 = nn.Sequential(_cb3(c, c * 2, 3, 2, 1), nn.ReLU(inplace=True))
        self.conv2 = _cb3(c * 2, c * 2, 3, 1, 1)
        self.conv3 = nn.Sequential(_cb3(c * 2, c * 2, 3, 2, 1), nn.ReLU(inplace=True))
        self.conv4 = nn.Sequential(_cb3(c * 2, c * 2, 3, 1, 1), nn.ReLU(inplace=True))
        self.conv5 = nn.Sequential(nn.ConvTranspose3d(c * 2, c * 2, 3, padding=1, output_padding=1, stride=2, bias=False), nn.BatchNorm3d(c * 2))
        self.conv6 = nn.Sequential(nn.ConvTranspose3d(c * 2, c, 3, padding=1, output_padding=1, stride=2, bias=False), nn.BatchNorm3d(c))

    def forward(self, x, presqu, postsqu):
        out = self.conv1(x)
        pre = self.conv2(out)
        pre = F.relu(pre + postsqu) if postsqu is not None else F.relu(pre)
        out = self.conv4(self.conv3(pre))
        post = F.relu(self.conv5(out) + (presqu if presqu is not None else pre))
        return self.conv6(post), pre, post


class PSMNet(nn.Module):
    """models_psmnet/stackhourglass.py:52-160 (device-agnostic: no hard-coded .cuda())."""

    def __init__(self, maxdisp):
        super().__init__()
        self.maxdisp = maxdisp
        self.feature_extraction = feature_extraction()
        self.dres0 = nn.Sequential(_cb3(64, 32, 3, 1, 1), nn.ReLU(inplace=True), _cb3(32, 32, 3, 1, 1), nn.ReLU(inplace=True))
        self.dres1 = nn.Sequential(_cb3(32, 32, 3, 1, 1), nn.ReLU(inplace=True), _cb3(32, 32, 3, 1, 1))
        self.dres2, self.dres3, self.dres4 = hourglass(32), hourglass(32), hourglass(32)
        for j in (1, 2, 3):
            setattr(self, 'classif%d' % j, nn.Sequential(_cb3(32, 32, 3, 1, 1), nn.ReLU(inplace=True), nn.Conv3d(32, 1, 3, padding=1, bias=False)))

    def _regress(self, cost, size):
        c = F.interpolate(cost, [self.maxdisp, size[0], size[1]], mode='trilinear').squeeze(1)
        p = F.softmax(c, 1)
        d = torch.arange(self.maxdisp, dtype=p.dtype, device=p.device).view(1, -1, 1, 1)
        return torch.sum(p * d, 1)

    def forward(self, left, right):
        ref, tgt = self.feature_extraction(left), self.feature_extraction(right)
        B, C, H, W = ref.shape
        D = self.maxdisp // 4
        cost = ref.new_zeros(B, C * 2, D, H, W)
        for i in range(D):
            if i > 0:
                cost[:, :C, i, :, i:] = ref[:, :, :, i:]
                cost[:, C:, i, :, i:] = tgt[:, :, :, :-i]
            else:
                cost[:, :C, i] = ref
                cost[:, C:, i] = tgt
        cost0 = self.dres0(cost)
        cost0 = self.dres1(cost0) + cost0
        out1, pre1, post1 = self.dres2(cost0, None, None)
        out1 = out1 + cost0
        out2, pre2, post2 = self.dres3(out1, pre1, post1)
        out2 = out2 + cost0
        out3, pre3, post3 = self.dres4(out2, pre1, post2)
        out3 = out3 + cost0
        cost1 = self.classif1(out1)
        cost2 = self.classif2(out2) + cost1
        cost3 = self.classif3(out3) + cost2
        pred3 = self._regress(cost3, left.shape[2:])
        if self.training:
            return self._regress(cost1, left.shape[2:]), self._regress(cost2, left.shape[2:]), pred3
        return pred3


# --------------------------------------------------------------------------- dsnet (models/dsnet_t2.py:119-321)
class dsnet(nn.Module):
    """models/dsnet_t2.py:119-321 — the line-for-line PyTorch port of the TF baseline_SDnet_small_fixed graph
    (2-D 17x17 correlation, stride-2 transposed convs in the segmentation decoder)."""
    _no_corr = False

    def __init__(self, CFG, labels=8, pretrained=False, backbone='densenet'):
        super().__init__()
        self.resnet_features = piramidNet(pretrained=pretrained)
        for j in (1, 2, 3):
            setattr(self, 'conv2d_ba%d' % j, _img_conv(3))
        self.correlation_sampler = SpatialCorrelationSampler(1, (17, 17), 1, 0, dilation_patch=1)
        self.corrConv2d = _c1x1(512 if self._no_corr else 289, 128)
        self.conv1d_1 = _c1x1(2048, 64)
        self.Conv2DownUp1 = Conv2DownUp(64, 32, 3)
        self.Conv2DownUp2 = nn.Sequential(Conv2DownUp(32, 32, 3, lastLayer=False), ConvTranspose2dSame(32, labels, 3, 1, padding='same', init_he=False))
        self.Conv2DownUp3 = Conv2DownUp(32, 128, 3)
        self.Conv2DownUp4 = Conv2DownUp(256, 64, 3)
        self.conv1d_2 = _c1x1(65, 64)
        self.Conv2DownUp5 = Conv2DownUp(64, 64, 5, lastLayer=False)
        self.dispoutConv = ConvTranspose2dSame(64, 1, 5, padding='same', init_he=False)
        self.conv1d_3 = _c1x1(96, 64)
        self.Conv2DownUp6 = Conv2DownUp(64, 64, 5)
        self.conv1d_4 = _c1x1(192, 64)
        self.conv2DT_BA1 = nn.Sequential(deconvbn(64, 32, 3, 2, 'same', 1), nn.ReLU(inplace=True))
        self.conv1d_5 = _c1x1(96, 32)
        self.conv2DT_BA2 = nn.Sequential(deconvbn(32, 32, 3, 2, 'same', 1), nn.ReLU(inplace=True))
        self.conv1d_6 = _c1x1(33, 32)
        self.Conv2DownUp7 = Conv2DownUp(32, 32, 5, lastLayer=False)
        self.branchConv = ConvTranspose2dSame(32, labels, 5, padding='same', init_he=False)
        self.conv1d_9 = _c1x1(448, 128)
        self.conv1d_7 = _c1x1(256, 128)
        self.Conv2DownUp8 = Conv2DownUp(32, 64, 3)
        self.Conv2DownUp9 = Conv2DownUp(256, 64, 3)
        self.conv1d_8 = _c1x1(65, 64)
        self.Conv2DownUp10 = nn.Sequential(Conv2DownUp(64, 64, 5, lastLayer=False), ConvTranspose2dSame(64, 1, 5, padding='same', init_he=False))

    def forward(self, input_a, input_b):
        a = self.resnet_features(input_a)   # a_0..a_4, pyramid(tap 2), pyramid(tap 0)
        b = self.resnet_features(input_b)
        size = input_a.shape[2:]
        xl3, xl2, xl1 = self.conv2d_ba3(input_a), self.conv2d_ba1(input_a), self.conv2d_ba2(input_a)
        x = F.interpolate(torch.cat((a[4], b[4]), 1), scale_factor=2, mode='nearest')
        x = self.Conv2DownUp1(self.conv1d_1(x))
        x1 = F.interpolate(x, scale_factor=2, mode='nearest')
        seg1 = F.interpolate(self.Conv2DownUp2(x1), scale_factor=8, mode='nearest')
        seg1 = F.log_softmax(F.interpolate(seg1, size=size, mode='bilinear'), 1)
        if self._no_corr:      # dsnetnoCorr (models/dsnet_t2.py:697-700): concatenated tap-2 maps instead of the correlation
            y = self.corrConv2d(torch.cat((a[2], b[2]), 1))
        else:
            y = self.correlation_sampler(a[5], b[5])
            n, ph, pw, h, w = y.shape
            y = self.corrConv2d(y.reshape(n, ph * pw, h, w) / a[5].size(1))
        y1 = F.interpolate(self.Conv2DownUp3(x1), size=y.shape[2:], mode='bilinear')
        y = self.Conv2DownUp4(torch.cat((y1, y), 1))
        y2 = F.interpolate(y, scale_factor=8)
        xl2 = F.interpolate(xl2, size=y2.shape[2:], mode='bilinear')
        d = self.dispoutConv(self.Conv2DownUp5(self.conv1d_2(torch.cat((y2, xl2), 1))))
        disp = F.interpolate(d, size=size, mode='bilinear')
        x = F.interpolate(x, scale_factor=4)
        y3 = F.interpolate(y, scale_factor=2)
        x = F.interpolate(x, y3.shape[2:], mode='bilinear')
        x = self.Conv2DownUp6(self.conv1d_3(torch.cat((x, y3), 1)))
        x = F.interpolate(x, a[1].shape[2:], mode='bilinear')
        x = self.conv2DT_BA1(self.conv1d_4(torch.cat((x, a[1]), 1)))
        x3 = x
        x = F.interpolate(x, a[0].shape[2:], mode='bilinear')
        x = self.conv2DT_BA2(self.conv1d_5(torch.cat((x, a[0]), 1)))
        xl1 = F.interpolate(xl1, x.shape[2:], mode='bilinear')
        s2 = self.branchConv(self.Conv2DownUp7(self.conv1d_6(torch.cat((x, xl1), 1))))
        s2 = F.interpolate(F.log_softmax(s2, 1), size, mode='bilinear')
        seg2 = 0.9 * s2 + 0.1 * seg1
        y4 = self.conv1d_9(torch.cat((a[6], b[6]), 1))
        y = F.interpolate(y, scale_factor=4)
        y = F.interpolate(y, y4.shape[2:], mode='bilinear')
        y = torch.cat((y4, y), 1)
        y5 = self.Conv2DownUp8(x3)
        y = F.interpolate(y, y5.shape[2:], mode='bilinear')
        y = self.Conv2DownUp9(torch.cat((y5, y), 1))
        y = F.interpolate(y, scale_factor=2)
        xl3 = F.interpolate(xl3, y.shape[2:], mode='bilinear')
        d2 = self.Conv2DownUp10(self.conv1d_8(torch.cat((y, xl3), 1)))
        d2 = F.interpolate(d2, size, mode='bilinear')
        return seg1, disp, seg2, 0.8 * d2 + 0.2 * disp


class dsnetnoCorr(dsnet):
    """models/dsnet_t2.py:620-823 — PyTorch port of the TF baseline_SDnet_small graph: dsnet without the correlation."""
    _no_corr = True

    def __init__(self, CFG, labels=8, pretrained=False):
        super().__init__(CFG, labels=labels, pretrained=pretrained)
