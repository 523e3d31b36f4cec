"""Drop-in nn.Module surface of the hot path (same class names, constructor
signatures and state_dict keys as the reference), every forward running on the
hand-written HIP kernels of libsdhip.so.
"""
import torch
import torch.nn as nn

from . import ops



class CFG:
    """The argparse fields the model constructors read (models/dsnet_t2.py:944-953: dropout, multaskloss, aspp, use_att,
    hanet, convDeconvOut, abilation) as a plain attribute bag, for callers that do not carry the reference's argparse
    namespace (bench.py, tools); any object with these attributes works."""

    def __init__(self, dropout=0.0, multaskloss=0, aspp=0, use_att=1, hanet=0, convDeconvOut=0, abilation=''):
        self.dropout, self.multaskloss, self.aspp, self.use_att = dropout, multaskloss, aspp, use_att
        self.hanet, self.convDeconvOut, self.abilation = hanet, convDeconvOut, abilation


class SpatialCorrelationSampler(nn.Module):
    """Replacement for `spatial_correlation_sampler.SpatialCorrelationSampler`
    as constructed at models/dsnet_t2.py:1078-1087: forward(input1, input2) ->
    (B, PH, PW, H, W), differentiable w.r.t. both inputs.  Only the
    configuration the reference uses is implemented (kernel_size=1, stride=1,
    padding=0, dilation=1); anything else raises, as the spec'd error behaviour."""

    def __init__(self, kernel_size=1, patch_size=1, stride=1, padding=0, dilation=1, dilation_patch=1):
        super().__init__()
        if (kernel_size, stride, padding, dilation) != (1, 1, 0, 1):
            raise NotImplementedError("sdhip SpatialCorrelationSampler supports kernel_size=1, stride=1, padding=0, "
                                      "dilation=1 (the only configuration on the reference's hot path)")
        self.patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.dilation_patch = int(dilation_patch)

    def forward(self, input1, input2):
        return ops.correlation(input1, input2, self.patch_size[0], self.patch_size[1], self.dilation_patch)


# --------------------------------------------------------------------------- conv building blocks
import math

import torch.nn.functional as F  # noqa: F401  (only for parameter containers / host-side shape helpers)


def _he_init(mods):
    """Weight init of convbn/deconvbn/conv2dSame (models/dsnet_t2.py:37-43, models/torch_model.py:260-266)."""
    for m in mods:
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            m.weight.data.normal_(0, math.sqrt(2.0 / (m.kernel_size[0] * m.kernel_size[1] * m.out_channels)))
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()


_dropout_ids = {}


def _dropout_id(mod):
    """Stable per-module stream id of a nn.Dropout (the mask is a function of (step seed, id, element index))."""
    return _dropout_ids.setdefault(id(mod), 1000 + len(_dropout_ids))


class conv2dSame(nn.Module):
    """models/torch_model.py:236-281. `c2d` is kept as the parameter container (state_dict key `c2d.weight`)."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding='valid', dilation=1, bias=False):
        super().__init__()
        self.padding = padding
        self.c2d = nn.Conv2d(in_channel, out_channel, kernel_size, stride=stride, dilation=dilation, bias=bias)
        _he_init([self.c2d])

    def _geom(self):
        c = self.c2d
        return dict(kind='conv', stride=c.stride[0], dilation=c.dilation[0], padding='same' if self.padding == 'same' else 0)

    def run(self, x, act=0):
        return ops.conv2d(x, self.c2d.weight, self.c2d.bias, act=act, **self._geom())

    def run_bn(self, x, bn, act=0, residual=None, groups=1, **slots):
        return ops.conv_bn_act(x, self.c2d.weight, bn, act=act, residual=residual, groups=groups, **slots, **self._geom())

    def forward(self, x):
        return self.run(x)


class ConvTranspose2dSame(nn.Module):
    """models/torch_model.py:284-349 (stride 1: a correlation with flipped, transposed weights — no (k-1) border
    is computed and thrown away, no crop copy)."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding='valid', dilation=1, bias=False, init_he=True):
        super().__init__()
        self.padding = padding
        self.ct2d = nn.ConvTranspose2d(in_channel, out_channel, kernel_size, stride=stride, dilation=dilation, bias=bias)

    def _geom(self):
        c = self.ct2d
        if self.padding != 'same' or c.dilation[0] != 1 and c.stride[0] != 1:
            raise NotImplementedError("ConvTranspose2dSame: only padding='same' is on the native path")
        return dict(kind='deconv', stride=1, dilation=c.dilation[0], padding='ctsame')

    def run(self, x, act=0):
        c = self.ct2d
        if c.stride[0] != 1:   # stride 2 (dsnet's conv2DT_BA*): stride-1 correlation over the zero-stuffed input
            return ops.deconv2d_strided(x, c.weight, c.bias, c.stride[0], act=act)
        return ops.conv2d(x, c.weight, c.bias, act=act, **self._geom())

    def run_bn(self, x, bn, act=0, residual=None, groups=1, **slots):
        c = self.ct2d
        if c.stride[0] != 1:
            if any(v is not None for v in slots.values()):
                raise NotImplementedError("gradient / BatchNorm slots are wired for the stride-1 blocks only")
            return ops.deconv2d_strided(x, c.weight, None, c.stride[0], bn=bn, act=act, residual=residual, groups=groups)
        return ops.conv_bn_act(x, c.weight, bn, act=act, residual=residual, groups=groups, **slots, **self._geom())

    def forward(self, x):
        return self.run(x)


class convbn(nn.Module):
    """models/dsnet_t2.py:16-46."""
    _conv = conv2dSame

    def __init__(self, in_channel, out_channel, kernel_size, stride, pad, dilation, batchnorm=True):
        super().__init__()
        seq = [self._conv(in_channel, out_channel, kernel_size, stride, pad, dilation, bias=not batchnorm)]
        if batchnorm:
            seq.append(nn.BatchNorm2d(out_channel))
        self.layers = nn.Sequential(*seq)
        _he_init(self.modules())

    def fused(self, x, act=0, residual=None, groups=1, **slots):
        """conv -> BatchNorm (batch statistics from the conv epilogue) -> activation (-> + residual): one autograd node.
        slots: in_slot / res_slot of ops.conv_bn_act."""
        conv = self.layers[0]
        if len(self.layers) == 1:
            if any(v is not None for v in slots.values()):
                raise NotImplementedError("gradient slots need the BatchNorm form of the block")
            y = conv.run(x, act=act)
            return y if residual is None else ops.affine_act(y, None, None, residual, 0)
        return conv.run_bn(x, self.layers[1], act=act, residual=residual, groups=groups, **slots)

    def forward(self, x):
        return self.fused(x)


class deconvbn(convbn):
    """models/dsnet_t2.py:48-77."""
    _conv = ConvTranspose2dSame


def _act_block(block, p=0.0):
    return nn.Sequential(block, nn.ReLU(inplace=True), nn.Dropout(p=p))


def run_act_block(seq, x, residual=None, groups=1, **slots):
    """Sequential(convbn|deconvbn, ReLU[, Dropout]) as one fused conv+BN+ReLU(+skip).  With Dropout(p > 0) in training
    mode (models/dsnet_t2.py:85-93; the shipped recipe has p = 0) the mask sits between the ReLU and the skip add, so the
    skip is added after the dropout kernel instead of inside the BatchNorm pass."""
    if len(seq) > 2 and seq[2].p and seq[2].training:
        y = ops.dropout(seq[0].fused(x, act=1, groups=groups), seq[2].p, True, _dropout_id(seq[2]))
        return y if residual is None else ops.add(y, residual)
    return seq[0].fused(x, act=1, residual=residual, groups=groups, **slots)


GRAD_SLOTS = not ops._lib.DIAG_NO_GRAD_SLOTS    # tests / diagnostics (SDHIP_DIAG_NO_GRAD_SLOTS)
BN_SLOTS = not ops._lib.DIAG_NO_BN_SLOTS        # ... (SDHIP_DIAG_NO_BN_SLOTS): BatchNorm backward reductions in the consumer's data gradient: False hands every skip gradient back to autograd (one elementwise add each)


class Conv2DownUp(nn.Module):
    """models/dsnet_t2.py:80-117: c1 -> c2 -> c3 -> d3 (+c2) -> d4 (+c1) [-> d5]; the skip adds are fused into the
    BatchNorm+ReLU pass of d3 / d4."""

    def __init__(self, in_channels, out_channels=3, kernel_size=3, lastLayer=True, dropout=0):
        super().__init__()
        self.lastLayer = lastLayer
        o, k = out_channels, kernel_size
        self.c1 = _act_block(convbn(in_channels, o, k, 1, 'same', 1), dropout)
        self.c2 = _act_block(convbn(o, o, k, 1, 'same', 1), dropout)
        self.c3 = _act_block(convbn(o, o, k, 1, 'same', 1), dropout)
        self.d3 = _act_block(deconvbn(o, o, k, 1, 'same', 1), dropout)
        self.d4 = _act_block(deconvbn(o, o, k, 1, 'same', 1), dropout)
        self.d5 = _act_block(deconvbn(o, o, k, 1, 'same', 1), dropout)

    def forward(self, x, groups=1):
        # x1 feeds c2 and (skip) d4, x2 feeds c3 and (skip) d3: in the backward pass the skip gradient is parked in a
        # GradSlot and the data gradient of c2 / c3 is accumulated onto it by the convolution launch itself (ops.GradSlot)
        slotted = GRAD_SLOTS and torch.is_grad_enabled() and x.is_cuda and len(self.c1[0].layers) > 1 and \
            not any(len(b) > 2 and b[2].p and b[2].training for b in (self.c2, self.c3, self.d3, self.d4))
        # d3's incoming gradient is d4's fresh data gradient (exclusive); d4's is d5's — or, without d5, whatever the rest
        # of the network hands in, which is only read
        s1 = ops.GradSlot(exclusive=self.lastLayer) if slotted else None
        s2 = ops.GradSlot(exclusive=True) if slotted else None
        # every layer's output has exactly one consumer inside the block (plus the slotted skips): that consumer's data
        # gradient takes the reductions of the layer's BatchNorm backward (ops.BNSlot)
        b = [ops.BNSlot() if slotted and BN_SLOTS else None for _ in range(5)]
        x1 = run_act_block(self.c1, x, groups=groups, out_bn=b[0])
        x2 = run_act_block(self.c2, x1, groups=groups, in_slot=s1, in_bn=b[0], out_bn=b[1])
        x = run_act_block(self.c3, x2, groups=groups, in_slot=s2, in_bn=b[1], out_bn=b[2])
        x = run_act_block(self.d3, x, residual=x2, groups=groups, res_slot=s2, in_bn=b[2], out_bn=b[3])
        x = run_act_block(self.d4, x, residual=x1, groups=groups, res_slot=s1, in_bn=b[3], out_bn=b[4] if self.lastLayer else None)
        return run_act_block(self.d5, x, groups=groups, in_bn=b[4]) if self.lastLayer else x


# --------------------------------------------------------------------------- pyramids, heads, full network
from .densenet import densenet121  # noqa: E402


def _pool_branch(p, cin):
    return nn.Sequential(nn.AvgPool2d(p, p), convbn(cin, 32, 3, 1, 'same', 1), nn.ReLU(inplace=True))


def _pyramid(branches, x, groups):
    """cat([x] + [bilinear_up(relu(bn(conv3x3(avgpool_p(x))))) for each branch]) (models/dsnet_t2.py:2037-2081).
    The pools share work: pool_2p = 2x2 pool of pool_p (identical windows, mean of equal-size means)."""
    order = sorted(range(len(branches)), key=lambda j: branches[j][0].kernel_size)
    outs, pooled, prev_p = [None] * len(branches), x, 1
    for j in order:
        p = branches[j][0].kernel_size
        p = p if isinstance(p, int) else p[0]
        if p % prev_p:
            pooled, prev_p = x, 1
        pooled = ops.avgpool(pooled, p // prev_p)
        prev_p = p
        y = branches[j][1].fused(pooled, act=1, groups=groups)
        outs[j] = ops.interpolate(y, size=x.shape[2:], mode='bilinear')
    return ops.concat([x] + outs)


class piramidNet2(nn.Module):
    """models/dsnet_t2.py:1893-2083 (densenet backbone)."""

    def __init__(self, pretrained=False, backbone='densenet'):
        super().__init__()
        if backbone != 'densenet':
            raise NotImplementedError("only the densenet backbone (the shipped recipe) is on the native path")
        self.backbone = backbone
        self.resnet_features = densenet121(pretrained)
        pv, cin = [128, 64, 32, 16, 8], [64, 128, 256]
        for j in range(5):
            setattr(self, 'branch0_%d' % j, _pool_branch(pv[j], cin[0]))
        for j in range(4):
            setattr(self, 'branch1_%d' % j, _pool_branch(pv[j + 1], cin[1]))
        for j in range(3):
            setattr(self, 'branch2_%d' % j, _pool_branch(pv[j + 2], cin[2]))

    def forward(self, x, groups=1):
        o = self.resnet_features(x, groups)
        b0 = _pyramid([getattr(self, 'branch0_%d' % j) for j in range(5)], o[0], groups)
        b1 = _pyramid([getattr(self, 'branch1_%d' % j) for j in range(4)], o[1], groups)
        b2 = _pyramid([getattr(self, 'branch2_%d' % j) for j in range(3)], o[2], groups)
        return o[0], o[1], o[2], o[3], o[4], b2, b1, b0


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
        x = ops.interpolate(x, scale_factor=2, mode='nearest')
        x = self.Conv2DownUp1(self.conv1d_1[0].run(x, act=1))
        x1 = ops.interpolate(x, scale_factor=2, mode='nearest')
        s = ops.upcat_conv1x1(x, xleft, self.conv1d_2[0].c2d.weight, act=1)
        if s is None:
            s = self.conv1d_2[0].run(ops.concat([ops.interpolate(x, size=xleft.shape[2:], mode='nearest'), xleft]), act=1)
        s = self.Conv2DownUp2[1](self.Conv2DownUp2[0](s))
        return x, x1, ops.interpolate(s, size=input_a.shape[2:], mode='nearest')


class minidsnetExt(nn.Module):
    """models/dsnet_t2.py:941-1299 — the network the shipped scripts train (`-net sdnet_mini_ext`), densenet backbone.
    forward(left, right) -> (seg_branch, disp_out, seg_branch2, disp_out).  Both towers run as one batch of two
    statistics groups (weights are shared; BatchNorm statistics stay per image side, as in the reference)."""

    def __init__(self, CFG, labels=8, pretrained=False, patch_type='', include_edges=False, backbone='densenet'):
        super().__init__()
        if backbone != 'densenet' or CFG.multaskloss:
            raise NotImplementedError("native path: densenet backbone, no multitask loss wrapper (models/dsnet_t2.py:1297) yet")
        self.include_edges = include_edges
        self.hanet = CFG.hanet
        dropout = CFG.dropout
        self.aspp_mod, self.use_att, self.convDeconvOut, self.abilation = CFG.aspp, CFG.use_att, CFG.convDeconvOut, CFG.abilation
        self.patch_type, self.backbone = patch_type, backbone
        feature_channel, inplane_seg2 = 1, 512
        if self.aspp_mod == 1:
            from .aspp import build_aspp
            self.aspp, inplane_seg2 = build_aspp('densenet_a1', 32), 256
        elif self.aspp_mod == 2:
            from .aspp import build_aspp
            self.aspp, inplane_seg2, feature_channel = build_aspp('densenet_a3', 32), 273, 64
        self.resnet_features = piramidNet2(pretrained, backbone)
        for j in range(4):   # the auxiliary image convolutions see the edge map as a 4th channel (models/dsnet_t2.py:1061-1069)
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
        if self.hanet:   # models/dsnet_t2.py:1135-1150
            from .hanet import HANet_Conv
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
        B, _, H, W = input_a.shape
        both, img_a = _stereo_buffer(input_a, input_b, self.include_edges)
        t = self.resnet_features(both, groups=2)          # taps of both towers, batch = [left | right]
        halves = [ops.split_batch(u, B) for u in t]   # left / right tower outputs (one gradient buffer per tap in the backward)
        a = [h[0] for h in halves]
        b = [h[1] for h in halves]
        xl3 = self.conv2d_ba3[0].fused(img_a, act=1)      # computed (and unused) exactly as in the reference
        xl2 = self.conv2d_ba1[0].fused(img_a, act=1)
        xl1 = self.conv2d_ba2[0].fused(img_a, act=1)
        xl0 = self.conv2d_ba0[0].fused(img_a, act=1)
        del xl3
        x, x1, seg1 = self.segNet(ops.concat([a[4], b[4]]), input_a, input_b, xl0)

        y = self.correlation_sampler(a[5], b[5])
        if self.patch_type == '1dcorr':
            y = torch.squeeze(y, 1)
            y = self.corrConv2d[0].run(y, act=1)
        else:
            n, ph, pw, h, w = y.shape
            y = y.reshape(n, ph * pw, h, w)
            # the reference divides the 2-D correlation by C (models/dsnet_t2.py:1193); fold it into the 1x1 conv input
            y = self.corrConv2d[0].run(ops.affine_act(y, _const(1.0 / a[5].size(1), ph * pw, y.device), None), act=1)
        y1 = self.Conv2DownUp3(a[5] if 'no_dec1' in self.abilation else x1)
        y1 = ops.interpolate(y1, size=y.shape[2:], mode='bilinear')
        y = self.Conv2DownUp4(ops.concat([y1, y]))
        up8 = (8 * y.shape[2], 8 * y.shape[3])
        xl2 = ops.interpolate(xl2, size=up8, mode='bilinear')
        d0 = ops.upcat_conv1x1(y, xl2, self.conv1d_2[0].c2d.weight, act=1)     # x8 upsample + concat folded into the 1x1 (bf16)
        if d0 is None:
            d0 = self.conv1d_2[0].run(ops.concat([ops.interpolate(y, scale_factor=8), xl2]), act=1)
        d = self.Conv2DownUp5(d0)
        disp = ops.interpolate(self.dispoutConv(d), size=input_a.shape[2:], mode='bilinear')

        if self.aspp_mod == 1:
            s2 = self.aspp(a[1])
        elif self.aspp_mod == 2:
            s2b = self.aspp(t[3], groups=2)
            s21, s22 = ops.split_batch(s2b, B)
            s2 = ops.concat([torch.squeeze(self.s2_corr_sampler(s21, s22), 1), s21])
        else:
            s2 = ops.concat([a[6], b[6]])
        s2 = self.Conv2DownUp6(self.conv1d_4[0].run(s2, act=1))
        y3 = ops.interpolate(y, size=s2.shape[2:])
        if 'no_dec3' not in self.abilation:
            x3 = ops.interpolate(self.Conv2DownUp8(x1), size=s2.shape[2:])
            if self.use_att:
                s2_d = self.Conv2DownUp7(ops.concat([s2, y3]))
                at_d = _seq_dropout(self.conv1d_at_d[2], self.conv1d_at_d[0].run(s2_d, act=2))
                s2_s = self.Conv2DownUp9(ops.concat([s2, x3]))
                at_s = _seq_dropout(self.conv1d_at_s[2], self.conv1d_at_s[0].run(s2_s, act=2))
                s2 = ops.concat([ops.mul_bcast(s2_d, at_s), ops.mul_bcast(s2_s, at_d)])
            else:
                s2 = ops.concat([s2, x3, y3])
        s2 = self.Conv2DownUp10(s2)
        if self.aspp_mod == 2:
            s2 = ops.concat([ops.interpolate(s2, size=a[0].shape[2:]), a[0]])
            seg2 = self.conv1d_5[0].run(s2, act=1)
            seg2 = self.Conv2DownUp11[1](self.Conv2DownUp11[0](seg2))
            seg2 = ops.interpolate(seg2, size=input_a.shape[2:], mode='nearest')
        else:
            seg2 = ops.upcat_conv1x1(s2, xl1, self.conv1d_5[0].c2d.weight, act=1)
            if seg2 is None:
                seg2 = self.conv1d_5[0].run(ops.concat([ops.interpolate(s2, size=xl1.shape[2:]), xl1]), act=1)
            seg2 = self.Conv2DownUp11[0](seg2)
            if self.convDeconvOut:
                s = self.convOutput2(seg2)
                seg2 = ops.affine_act(self.convOutput(seg2), None, None, s) if self.convDeconvOut == 2 else s
            else:
                seg2 = self.Conv2DownUp11[1](seg2)
            if self.hanet:   # only on this branch, as upstream (models/dsnet_t2.py:1287-1289): with aspp == 2 the head is built but unused
                seg2, _ = self.hanet_last(a[0], seg2, pos, attention_loss=True)
        return seg1, disp, seg2, disp


def _stereo_buffer(input_a, input_b, include_edges=False):
    """Both images in one NHWC buffer, channels zero-padded to 8: one pixel = one 16-byte chunk, so the image convolutions
    (conv0 7x7/2, conv2d_ba* 5x5 dil 2) stage their input with vector loads; the padded weight columns are zero.
    With include_edges the inputs carry the edge map as a 4th channel (models/dsnet_t2.py:1153-1158): the towers read the
    first three (their weight has three input channels — the 4th meets zero weight columns), the auxiliary convolutions of
    the LEFT image all four.  Returns (both towers' batch, left image view)."""
    B, Cimg, H, W = input_a.shape
    n = 4 if include_edges else 3
    if Cimg != n:
        raise ValueError("expected %d-channel images (include_edges=%s), got %d" % (n, include_edges, Cimg))
    both8 = torch.zeros((2 * B, H, W, 8), dtype=input_a.dtype, device=input_a.device)
    both8[:B, :, :, :n] = input_a.permute(0, 2, 3, 1)
    both8[B:, :, :, :3] = input_b[:, :3].permute(0, 2, 3, 1)
    both = both8.permute(0, 3, 1, 2)
    return both, both[:B]


def _seq_dropout(mod, x):
    return ops.dropout(x, mod.p, mod.training, _dropout_id(mod)) if mod.p else x


def _const(v, n, device):
    return torch.full((1, n), v, dtype=torch.float32, device=device)


class piramidNet(nn.Module):
    """models/dsnet_t2.py:324-390 (the pyramid of `dsnet`): taps + pyramid over tap 2 + pyramid over tap 0."""

    def __init__(self, pretrained=False):
        super().__init__()
        self.resnet_features = densenet121(pretrained)
        pv = [128, 64, 32, 16, 8]
        for j in range(5):
            setattr(self, 'branch0_%d' % j, _pool_branch(pv[j], 64))
        for j in range(3):
            setattr(self, 'branch1_%d' % j, _pool_branch(pv[j + 2], 256))

    def forward(self, x, groups=1):
        o = self.resnet_features(x, groups)
        b0 = _pyramid([getattr(self, 'branch0_%d' % j) for j in range(5)], o[0], groups)
        b2 = _pyramid([getattr(self, 'branch1_%d' % j) for j in range(3)], o[2], groups)
        return o[0], o[1], o[2], o[3], o[4], b2, b0


class minidsnet(nn.Module):
    """models/dsnet_t2.py:825-913 (`-net sdnet_mini`, util/utilLoadNetwork.py:10): the first half of minidsnetExt over the
    `piramidNet` pyramid — coarse segmentation head + correlation-based disparity head.
    forward(left, right) -> (seg_branch, disp_out, seg_branch, disp_out)."""

    def __init__(self, CFG, labels=8, pretrained=False, patch_type='', include_edges=False, backbone='densenet'):
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
        self.conv1d_3 = _c1x1(96, 64)          # never called upstream either: kept for the state_dict

    def forward(self, input_a, input_b):
        B = input_a.shape[0]
        both, img_a = _stereo_buffer(input_a, input_b, self.include_edges)
        t = self.resnet_features(both, groups=2)
        halves = [ops.split_batch(u, B) for u in t]
        a = [h[0] for h in halves]
        b = [h[1] for h in halves]
        xl3 = self.conv2d_ba3[0].fused(img_a, act=1)      # computed (and unused) exactly as in the reference, like xl1
        xl2 = self.conv2d_ba1[0].fused(img_a, act=1)
        xl1 = self.conv2d_ba2[0].fused(img_a, act=1)
        xl0 = self.conv2d_ba0[0].fused(img_a, act=1)
        del xl3, xl1
        x, x1, seg1 = self.segNet(ops.concat([a[4], b[4]]), input_a, input_b, xl0)
        y = self.correlation_sampler(a[5], b[5])
        if self.patch_type == '1dcorr':
            y = self.corrConv2d[0].run(torch.squeeze(y, 1), act=1)
        else:
            n, ph, pw, h, w = y.shape
            y = self.corrConv2d[0].run(ops.affine_act(y.reshape(n, ph * pw, h, w), _const(1.0 / a[5].size(1), ph * pw, y.device), None), act=1)
        y1 = ops.interpolate(self.Conv2DownUp3(x1), size=y.shape[2:], mode='bilinear')
        y = self.Conv2DownUp4(ops.concat([y1, y]))
        xl2 = ops.interpolate(xl2, size=(8 * y.shape[2], 8 * y.shape[3]), mode='bilinear')
        d0 = ops.upcat_conv1x1(y, xl2, self.conv1d_2[0].c2d.weight, act=1)
        if d0 is None:
            d0 = self.conv1d_2[0].run(ops.concat([ops.interpolate(y, scale_factor=8), xl2]), act=1)
        disp = ops.interpolate(self.dispoutConv(self.Conv2DownUp5(d0)), size=input_a.shape[2:], mode='bilinear')
        return seg1, disp, seg1, disp


class dsnet(nn.Module):
    """models/dsnet_t2.py:119-321 — the PyTorch port of the TF `baseline_SDnet_small_fixed` graph (BASELINE config 2):
    2-D 17x17 correlation, stride-2 transposed convs, log-softmax heads blended 0.9/0.1 and 0.8/0.2.
    forward(left, right) -> (seg_branch, disp_out, seg_branch2, disp_out2)."""
    _match_channels = 289      # input channels of corrConv2d: the 17 x 17 displacements

    def __init__(self, CFG, labels=8, pretrained=False, backbone='densenet'):
        super().__init__()
        self.resnet_features = piramidNet(pretrained=pretrained)
        for j in (1, 2, 3):
            setattr(self, 'conv2d_ba%d' % j, _img_conv(3))
        self.correlation_sampler = SpatialCorrelationSampler(1, (17, 17), 1, 0, dilation_patch=1)
        self.corrConv2d = _c1x1(self._match_channels, 128)
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
        B, _, H, W = input_a.shape
        both8 = torch.zeros((2 * B, H, W, 8), dtype=input_a.dtype, device=input_a.device)
        both8[:B, :, :, :3] = input_a.permute(0, 2, 3, 1)
        both8[B:, :, :, :3] = input_b.permute(0, 2, 3, 1)
        both = both8.permute(0, 3, 1, 2)
        img_a = both[:B]
        t = self.resnet_features(both, groups=2)
        halves = [ops.split_batch(u, B) for u in t]   # left / right tower outputs (one gradient buffer per tap in the backward)
        a = [h[0] for h in halves]
        b = [h[1] for h in halves]
        size = (H, W)
        up = ops.interpolate
        xl3 = self.conv2d_ba3[0].fused(img_a, act=1)
        xl2 = self.conv2d_ba1[0].fused(img_a, act=1)
        xl1 = self.conv2d_ba2[0].fused(img_a, act=1)
        x = up(ops.concat([a[4], b[4]]), scale_factor=2, mode='nearest')
        x = self.Conv2DownUp1(self.conv1d_1[0].run(x, act=1))
        x1 = up(x, scale_factor=2, mode='nearest')
        seg1 = up(self.Conv2DownUp2[1](self.Conv2DownUp2[0](x1)), scale_factor=8, mode='nearest')
        seg1 = ops.log_softmax(up(seg1, size=size, mode='bilinear'))
        y = self._match(a, b)
        y1 = up(self.Conv2DownUp3(x1), size=y.shape[2:], mode='bilinear')
        y = self.Conv2DownUp4(ops.concat([y1, y]))
        xl2 = up(xl2, size=(8 * y.shape[2], 8 * y.shape[3]), mode='bilinear')
        d0 = ops.upcat_conv1x1(y, xl2, self.conv1d_2[0].c2d.weight, act=1)
        if d0 is None:
            d0 = self.conv1d_2[0].run(ops.concat([up(y, scale_factor=8), xl2]), act=1)
        d = self.dispoutConv(self.Conv2DownUp5(d0))
        disp = up(d, size=size, mode='bilinear')
        x = up(x, scale_factor=4)
        y3 = up(y, scale_factor=2)
        x = up(x, size=y3.shape[2:], mode='bilinear')
        x = self.Conv2DownUp6(self.conv1d_3[0].run(ops.concat([x, y3]), act=1))
        x = up(x, size=a[1].shape[2:], mode='bilinear')
        x = self.conv2DT_BA1[0].fused(self.conv1d_4[0].run(ops.concat([x, a[1]]), act=1), act=1)
        x3 = x
        x = up(x, size=a[0].shape[2:], mode='bilinear')
        x = self.conv2DT_BA2[0].fused(self.conv1d_5[0].run(ops.concat([x, a[0]]), act=1), act=1)
        xl1 = up(xl1, size=x.shape[2:], mode='bilinear')
        s0 = ops.upcat_conv1x1(x, xl1, self.conv1d_6[0].c2d.weight, act=1)
        if s0 is None:
            s0 = self.conv1d_6[0].run(ops.concat([x, xl1]), act=1)
        s2 = self.branchConv(self.Conv2DownUp7(s0))
        s2 = up(ops.log_softmax(s2), size=size, mode='bilinear')
        seg2 = ops.axpby(0.9, s2, 0.1, seg1)
        y4 = self.conv1d_9[0].run(ops.concat([a[6], b[6]]), act=1)
        y = up(y, scale_factor=4)
        y = up(y, size=y4.shape[2:], mode='bilinear')
        y = ops.concat([y4, y])
        y5 = self.Conv2DownUp8(x3)
        y = up(y, size=y5.shape[2:], mode='bilinear')
        y = self.Conv2DownUp9(ops.concat([y5, y]))
        xl3 = up(xl3, size=(2 * y.shape[2], 2 * y.shape[3]), mode='bilinear')
        e0 = ops.upcat_conv1x1(y, xl3, self.conv1d_8[0].c2d.weight, act=1)
        if e0 is None:
            e0 = self.conv1d_8[0].run(ops.concat([up(y, scale_factor=2), xl3]), act=1)
        d2 = self.Conv2DownUp10[1](self.Conv2DownUp10[0](e0))
        d2 = up(d2, size=size, mode='bilinear')
        return seg1, disp, seg2, ops.axpby(0.8, d2, 0.2, disp)

    def _match(self, a, b):
        """Left/right matching features at 1/8 resolution: 2-D correlation of the tap-2 pyramids / C, then 1x1 + ReLU
        (models/dsnet_t2.py:221-224)."""
        y = self.correlation_sampler(a[5], b[5])
        n, ph, pw, h, w = y.shape
        y = y.reshape(n, ph * pw, h, w)
        return self.corrConv2d[0].run(ops.affine_act(y, ops._const_vec(1.0 / a[5].size(1), ph * pw, y.device), None), act=1)


class dsnetnoCorr(dsnet):
    """models/dsnet_t2.py:620-823 — the PyTorch port of the TF `baseline_SDnet_small` graph (BASELINE config 1): `dsnet`
    with the correlation replaced by a plain concatenation of the two towers' tap-2 maps (`:697-700`; the sampler is still
    constructed, `:630-634`, and has no parameters).  forward(left, right) -> (seg_branch, disp_out, seg_branch2, disp_out2)."""
    _match_channels = 512

    def __init__(self, CFG, labels=8, pretrained=False):
        super().__init__(CFG, labels=labels, pretrained=pretrained)

    def _match(self, a, b):
        return self.corrConv2d[0].run(ops.concat([a[2], b[2]]), act=1)
