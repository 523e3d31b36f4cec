"""PSMNet (stacked hourglass) of the reference (models_psmnet/stackhourglass.py, submodule.py), MI355X-native.

Same module tree / state_dict keys.  Volumes are (B*D, C, H, W) NHWC image stacks ([B][D][H][W][C] memory); the 3-D
convolutions fold their depth taps into the channel-chunk loop of the 2-D direct-conv kernels, the stride-2 transposed
convolutions run over zero-stuffed volumes, the concatenation cost volume is written by one kernel and the
upsample -> softmax -> regression tail is one fused kernel that never materialises the (B,192,H,W) tensor.
"""
import math

import torch
import torch.nn as nn

from . import ops


def convbn(in_planes, out_planes, kernel_size, stride, pad, dilation):
    """models_psmnet/submodule.py:10-13 (parameter container; run through _run2d)."""
    return nn.Sequential(nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride,
                                   padding=dilation if dilation > 1 else pad, dilation=dilation, bias=False),
                         nn.BatchNorm2d(out_planes))


def convbn_3d(in_planes, out_planes, kernel_size, stride, pad):
    """models_psmnet/submodule.py:16-19."""
    return nn.Sequential(nn.Conv3d(in_planes, out_planes, kernel_size=kernel_size, padding=pad, stride=stride, bias=False),
                         nn.BatchNorm3d(out_planes))


def _run2d(seq, x, act=0, residual=None, groups=1, in_slot=None, res_slot=None):
    c = seq[0]
    return ops.conv_bn_act(x, c.weight, seq[1], kind='conv', stride=c.stride[0], dilation=c.dilation[0], padding=c.padding[0],
                           act=act, residual=residual, groups=groups, in_slot=in_slot, res_slot=res_slot)


def _run3d(seq, x, D, act=0, residual=None, groups=1, in_slot=None, res_slot=None):
    c = seq[0]
    return ops.conv3d_bn_act(x, D, c.weight, seq[1], stride=c.stride[0], padding=c.padding[0], act=act, residual=residual,
                             groups=groups, in_slot=in_slot, res_slot=res_slot)


class BasicBlock(nn.Module):
    """models_psmnet/submodule.py:21-46."""
    expansion = 1

    def __init__(self, inplanes, planes, stride, downsample, pad, dilation):
        super().__init__()
        self.conv1 = nn.Sequential(convbn(inplanes, planes, 3, stride, pad, dilation), nn.ReLU(inplace=True))
        self.conv2 = convbn(planes, planes, 3, 1, pad, dilation)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x, groups=1):
        # x feeds conv1 and (identity blocks) the skip add: in the backward pass conv2's node parks the skip gradient — its own
        # incoming gradient, the fresh data gradient of whatever consumes the block — and conv1's data gradient is accumulated
        # onto it by the convolution launch (ops.GradSlot) instead of autograd adding two full maps (22 blocks per step)
        slot = ops.GradSlot(exclusive=True) if (self.downsample is None and torch.is_grad_enabled() and x.is_cuda and x.requires_grad) else None
        out = _run2d(self.conv1[0], x, act=1, groups=groups, in_slot=slot)
        skip = _run2d(self.downsample, x, groups=groups) if self.downsample is not None else x
        return _run2d(self.conv2, out, act=0, residual=skip, groups=groups, res_slot=slot)     # BN then += x, no ReLU


class disparityregression(nn.Module):
    """models_psmnet/submodule.py:56-64 — kept for the module surface; PSMNet.forward uses the fused soft-argmin."""

    def __init__(self, maxdisp):
        super().__init__()
        self.maxdisp = maxdisp


class feature_extraction(nn.Module):
    """models_psmnet/submodule.py:66-141."""

    def __init__(self):
        super().__init__()
        self.inplanes = 32
        self.firstconv = nn.Sequential(convbn(3, 32, 3, 2, 1, 1), nn.ReLU(inplace=True),
                                       convbn(32, 32, 3, 1, 1, 1), nn.ReLU(inplace=True),
                                       convbn(32, 32, 3, 1, 1, 1), nn.ReLU(inplace=True))
        self.layer1 = self._make_layer(BasicBlock, 32, 3, 1, 1, 1)
        self.layer2 = self._make_layer(BasicBlock, 64, 16, 2, 1, 1)
        self.layer3 = self._make_layer(BasicBlock, 128, 3, 1, 1, 1)
        self.layer4 = self._make_layer(BasicBlock, 128, 3, 1, 1, 2)
        for j, p in enumerate((64, 32, 16, 8)):
            setattr(self, 'branch%d' % (j + 1), nn.Sequential(nn.AvgPool2d((p, p), stride=(p, p)), convbn(128, 32, 1, 1, 0, 1),
                                                              nn.ReLU(inplace=True)))
        self.lastconv = nn.Sequential(convbn(320, 128, 3, 1, 1, 1), nn.ReLU(inplace=True),
                                      nn.Conv2d(128, 32, kernel_size=1, padding=0, stride=1, bias=False))

    def _make_layer(self, block, planes, blocks, stride, pad, dilation):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, pad, dilation)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, 1, None, pad, dilation))
        return nn.Sequential(*layers)

    def forward(self, x, groups=1):
        f = self.firstconv
        out = _run2d(f[0], x, act=1, groups=groups)
        out = _run2d(f[2], out, act=1, groups=groups)
        out = _run2d(f[4], out, act=1, groups=groups)
        for blk in self.layer1:
            out = blk(out, groups)
        raw = out
        for blk in self.layer2:
            raw = blk(raw, groups)
        out = raw
        for blk in self.layer3:
            out = blk(out, groups)
        skip = out
        for blk in self.layer4:
            skip = blk(skip, groups)
        # SPP: the four pools share work (pool_2p = 2x2 pool of pool_p), finest first
        feats, pooled, prev = {}, skip, 1
        for j in (4, 3, 2, 1):
            br = getattr(self, 'branch%d' % j)
            p = br[0].kernel_size[0]
            pooled = ops.avgpool(pooled, p // prev)
            prev = p
            y = _run2d(br[1], pooled, act=1, groups=groups)
            feats[j] = ops.interpolate(y, size=skip.shape[2:], mode='bilinear')
        cat = ops.concat([raw, skip, feats[4], feats[3], feats[2], feats[1]])
        y = _run2d(self.lastconv[0], cat, act=1, groups=groups)
        return ops.conv2d(y, self.lastconv[2].weight, None, kind='conv', padding=0)


class hourglass(nn.Module):
    """models_psmnet/stackhourglass.py:10-50."""

    def __init__(self, inplanes):
        super().__init__()
        self.conv1 = nn.Sequential(convbn_3d(inplanes, inplanes * 2, kernel_size=3, stride=2, pad=1), nn.ReLU(inplace=True))
        self.conv2 = convbn_3d(inplanes * 2, inplanes * 2, kernel_size=3, stride=1, pad=1)
        self.conv3 = nn.Sequential(convbn_3d(inplanes * 2, inplanes * 2, kernel_size=3, stride=2, pad=1), nn.ReLU(inplace=True))
        self.conv4 = nn.Sequential(convbn_3d(inplanes * 2, inplanes * 2, kernel_size=3, stride=1, pad=1), nn.ReLU(inplace=True))
        self.conv5 = nn.Sequential(nn.ConvTranspose3d(inplanes * 2, inplanes * 2, kernel_size=3, padding=1, output_padding=1, stride=2, bias=False),
                                   nn.BatchNorm3d(inplanes * 2))
        self.conv6 = nn.Sequential(nn.ConvTranspose3d(inplanes * 2, inplanes, kernel_size=3, padding=1, output_padding=1, stride=2, bias=False),
                                   nn.BatchNorm3d(inplanes))

    def forward(self, x, D, presqu, postsqu, add_out=None):
        """x: (B*D, C, H, W).  Returns (out [+ add_out], pre, post) like the reference (the `+ cost0` of
        stackhourglass.py:125,128,131 is fused into conv6's BatchNorm pass via add_out)."""
        out, D2 = _run3d(self.conv1[0], x, D, act=1)
        if postsqu is not None:
            pre, _ = _run3d(self.conv2, out, D2, act=0, residual=postsqu)
            pre = ops.relu(pre)
        else:
            pre, _ = _run3d(self.conv2, out, D2, act=1)
        out, D4 = _run3d(self.conv3[0], pre, D2, act=1)
        out, _ = _run3d(self.conv4[0], out, D4, act=1)
        post, _ = ops.deconv3d_s2_bn_act(out, D4, self.conv5[0].weight, self.conv5[1], act=0,
                                         residual=presqu if presqu is not None else pre)
        post = ops.relu(post)
        out, _ = ops.deconv3d_s2_bn_act(post, D2, self.conv6[0].weight, self.conv6[1], act=0, residual=add_out)
        return out, pre, post


class PSMNet(nn.Module):
    """models_psmnet/stackhourglass.py:52-160.  forward(left, right) -> (pred1, pred2, pred3) in training mode, pred3 in
    eval mode, each (B, H, W)."""

    def __init__(self, maxdisp):
        super().__init__()
        self.maxdisp = maxdisp
        self.feature_extraction = feature_extraction()
        self.dres0 = nn.Sequential(convbn_3d(64, 32, 3, 1, 1), nn.ReLU(inplace=True), convbn_3d(32, 32, 3, 1, 1), nn.ReLU(inplace=True))
        self.dres1 = nn.Sequential(convbn_3d(32, 32, 3, 1, 1), nn.ReLU(inplace=True), convbn_3d(32, 32, 3, 1, 1))
        self.dres2 = hourglass(32)
        self.dres3 = hourglass(32)
        self.dres4 = hourglass(32)
        for j in (1, 2, 3):
            setattr(self, 'classif%d' % j, nn.Sequential(convbn_3d(32, 32, 3, 1, 1), nn.ReLU(inplace=True),
                                                         nn.Conv3d(32, 1, kernel_size=3, padding=1, stride=1, bias=False)))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, math.sqrt(2. / (m.kernel_size[0] * m.kernel_size[1] * m.out_channels)))
            elif isinstance(m, nn.Conv3d):
                m.weight.data.normal_(0, math.sqrt(2. / (m.kernel_size[0] * m.kernel_size[1] * m.kernel_size[2] * m.out_channels)))
            elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _classify(self, seq, x, D):
        y, _ = _run3d(seq[0], x, D, act=1)
        return ops.conv3d(y, D, seq[2].weight, 1, 1)[0]

    def forward(self, left, right):
        B, _, H, W = left.shape
        both = torch.cat([left, right], 0)
        fea = self.feature_extraction(both, groups=2)        # both towers: one pass, two statistics groups
        ref, tgt = ops.split_batch(fea, B)
        D = self.maxdisp // 4
        cost = ops.cost_volume(ref, tgt, D)                   # (B*D, 64, H/4, W/4)
        c0, _ = _run3d(self.dres0[0], cost, D, act=1)
        c0, _ = _run3d(self.dres0[2], c0, D, act=1)
        # c0 feeds dres1[0] and the skip add behind dres1[2] (stackhourglass.py:122-123): the skip gradient is parked and dres1[0]'s
        # data gradient accumulates onto it (ops.GradSlot) instead of autograd adding two 32-channel volumes
        slot = ops.GradSlot(exclusive=True) if (torch.is_grad_enabled() and c0.requires_grad) else None
        c1, _ = _run3d(self.dres1[0], c0, D, act=1, in_slot=slot)
        cost0, _ = _run3d(self.dres1[2], c1, D, act=0, residual=c0, res_slot=slot)
        out1, pre1, post1 = self.dres2(cost0, D, None, None, add_out=cost0)
        out2, pre2, post2 = self.dres3(out1, D, pre1, post1, add_out=cost0)
        out3, pre3, post3 = self.dres4(out2, D, pre1, post2, add_out=cost0)   # pre1, as in the reference (:130)
        cost1 = self._classify(self.classif1, out1, D)
        cost2 = ops.add(self._classify(self.classif2, out2, D), cost1)
        cost3 = ops.add(self._classify(self.classif3, out3, D), cost2)
        pred3 = ops.soft_argmin(cost3, D, self.maxdisp, H, W)
        if self.training:
            return ops.soft_argmin(cost1, D, self.maxdisp, H, W), ops.soft_argmin(cost2, D, self.maxdisp, H, W), pred3
        return pred3
