"""DenseNet-121 feature tower of the reference (models/densenet.py), MI355X-native.

Same module tree / state_dict keys as the reference (`conv0`, `features.norm0`,
`denseblock.{0..6}.denselayerN.{norm1,conv1,norm2,conv2}`, `norm5`, `classifier`),
but a dense block runs as ONE autograd node over a pre-allocated NHWC channel
slab:

  * no torch.cat: every layer's 3x3 conv writes its 32 new channels straight into
    its channel slice of the slab (the reference re-concatenates all previous
    features for every layer, models/densenet.py:41-45,110-116);
  * no stand-alone BatchNorm pass: norm1/norm2 + ReLU are the fused input
    prologue of the 1x1 / 3x3 conv kernels, and the batch statistics they need
    are summed in the epilogue of the conv that produced the channels.  All
    layers that read a slab channel share its statistics (they only differ in
    gamma/beta), so each channel's sum / sum-of-squares is computed once;
  * left and right image run through the tower in one launch sequence as two
    statistics groups, which reproduces the reference's two separate calls
    (separate batch statistics, running stats updated left then right).
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from . import ops
from ._lib import call, dtype_code, ptr, stream_ptr


def _finalize(stats, bn, count, groups, training):
    """scale, shift, mean, invstd of one BatchNorm2d over (a slice of) a statistics slab."""
    C = bn.num_features
    dev = bn.weight.device
    out = [torch.empty((groups, C), dtype=torch.float32, device=dev) for _ in range(4)]
    if training:
        mom = 0.1 if bn.momentum is None else bn.momentum
        if bn.num_batches_tracked is not None:
            bn.num_batches_tracked += groups
        call("sdhip_bn_finalize", ptr(stats), stats.stride(1), ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean),
             ptr(bn.running_var), ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]), C, groups, float(count),
             float(bn.eps), float(mom), stream_ptr())
    else:
        call("sdhip_bn_finalize", None, 0, ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var),
             ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]), C, groups, float(count), float(bn.eps), 0.0, stream_ptr())
    return out


def _wgrad(x, ldx, dy, lddy, weight, B, H, W, Cin, Cout, k, pad, in_scale, in_shift, groups, dt):
    """dW of a stride-1 nn.Conv2d (weight (Cout,Cin,k,k)) whose input is prologue(x)."""
    T = k * k
    acc = torch.empty(ops._lib.packed_elems(Cout, Cin, T, dt), dtype=torch.float32, device=weight.device)
    call("sdhip_conv2d_wgrad", ptr(x), ptr(dy), ptr(acc), None, ptr(in_scale), ptr(in_shift), B, H, W, Cin, ldx,
         H, W, Cout, lddy, k, k, 1, 1, pad, pad, 1, groups, dt, stream_ptr())
    gw = torch.empty_like(weight, memory_format=torch.contiguous_format)
    call("sdhip_conv_unpack_wgrad", ptr(acc), ptr(gw), Cout, Cin, T, Cin * T, T, 0, 0, dt, stream_ptr())
    return gw


class _DenseBlockFn(torch.autograd.Function):
    """(slab, slab statistics) = dense_block(x0); see the module docstring."""

    @staticmethod
    def forward(ctx, x0, block, groups, *params):
        layers = list(block.values())
        L = len(layers)
        B, C0, H, W = x0.shape
        growth = layers[0].conv2.out_channels
        mid = layers[0].conv1.out_channels
        Ct = C0 + L * growth
        dev, dtype = x0.device, x0.dtype
        dt = dtype_code(x0)
        training = block.training
        npix = B * H * W
        count = (B // groups) * H * W
        slab = ops.empty_nhwc(B, Ct, H, W, dtype, dev)
        xv, ldx = ops.nhwc_view(x0)
        call("sdhip_affine_act", ptr(xv), ldx, ptr(slab), Ct, None, 0, None, None, npix, C0, 1, 0, dt, stream_ptr())
        S = torch.zeros((groups, 2, Ct), dtype=torch.float64, device=dev)
        if training:
            call("sdhip_channel_stats", ptr(slab), Ct, ptr(S), Ct, npix, C0, groups, 0, dt, stream_ptr())
        saved = []
        for li, layer in enumerate(layers):
            Cin = C0 + li * growth
            sc1, sh1, mu1, iv1 = _finalize(S[:, :, :Cin], layer.norm1, count, groups, training)
            w1 = ops.packed_weight(layer.conv1.weight, 'conv', 'fwd', dtype)
            y1 = ops.empty_nhwc(B, mid, H, W, dtype, dev)
            S2 = torch.zeros((groups, 2, mid), dtype=torch.float64, device=dev) if training else None
            ops._conv_launch(slab, Ct, w1, y1, mid, None, sc1, sh1, S2, B, H, W, Cin, H, W, mid, 1, 1, 1, 1, 0, 0,
                             True, groups, 0, False)
            sc2, sh2, mu2, iv2 = _finalize(S2, layer.norm2, count, groups, training)
            w2 = ops.packed_weight(layer.conv2.weight, 'conv', 'fwd', dtype)
            ops._conv_launch(y1, mid, w2, slab[:, Cin:Cin + growth], Ct, None, sc2, sh2,
                             S[:, :, Cin:Cin + growth] if training else None, B, H, W, mid, H, W, growth, 3, 3, 1, 1, 1, 1,
                             True, groups, 0, False)
            saved.append((y1, sc1, sh1, mu1, iv1, sc2, sh2, mu2, iv2))
        ctx.block, ctx.groups, ctx.saved, ctx.slab = block, groups, saved, slab
        ctx.geom = (B, C0, H, W, growth, mid, Ct, training, count)
        return slab, S

    @staticmethod
    def backward(ctx, g_slab_in, gS_in):
        layers = list(ctx.block.values())
        groups, slab = ctx.groups, ctx.slab
        B, C0, H, W, growth, mid, Ct, training, count = ctx.geom
        dev, dtype = slab.device, slab.dtype
        dt = dtype_code(slab)
        npix = B * H * W
        st = stream_ptr()
        # accumulators: gradient w.r.t. the slab and w.r.t. its statistics
        g_slab = ops.empty_nhwc(B, Ct, H, W, dtype, dev)
        gv, ldg = ops.nhwc_view(g_slab_in)
        call("sdhip_affine_act", ptr(gv), ldg, ptr(g_slab), Ct, None, 0, None, None, npix, Ct, 1, 0, dt, st)
        dS = gS_in.clone() if gS_in is not None else torch.zeros((groups, 2, Ct), dtype=torch.float64, device=dev)
        grads = []
        for li in range(len(layers) - 1, -1, -1):
            layer = layers[li]
            y1, sc1, sh1, mu1, iv1, sc2, sh2, mu2, iv2 = ctx.saved[li]
            Cin = C0 + li * growth
            sl_g = g_slab[:, Cin:Cin + growth]
            sl_x = slab[:, Cin:Cin + growth]
            # (1) total gradient of this layer's 32 output channels (all consumers are already accumulated)
            dy2 = ops.empty_nhwc(B, growth, H, W, dtype, dev)
            call("sdhip_stats_fix", ptr(sl_g), Ct, ptr(sl_x), Ct, ptr(dy2), growth, ptr(dS[:, :, Cin:Cin + growth]), Ct,
                 npix, growth, groups, dt, st)
            # (2) conv2 (3x3): data gradient w.r.t. relu(norm2(y1)), weight gradient
            wd2 = ops.packed_weight(layer.conv2.weight, 'conv', 'dgrad', dtype)
            gp2 = ops.empty_nhwc(B, mid, H, W, dtype, dev)
            ops._conv_launch(dy2, growth, wd2, gp2, mid, None, None, None, None, B, H, W, growth, H, W, mid, 3, 3, 1, 1, 1, 1,
                             False, 1, 0, False)
            gw2 = _wgrad(y1, mid, dy2, growth, layer.conv2.weight, B, H, W, mid, growth, 3, 1, sc2, sh2, groups, dt)
            # (3) through relu + norm2's affine, (4) norm2's statistics, (5) into y1
            dsc2 = torch.empty_like(sc2); dsh2 = torch.empty_like(sh2)
            call("sdhip_affine_act_bwd", ptr(gp2), mid, ptr(y1), mid, ptr(gp2), mid, ptr(sc2), ptr(sh2), ptr(dsc2), ptr(dsh2),
                 npix, mid, groups, 1, 0, dt, st)
            dg2 = torch.empty(mid, dtype=torch.float32, device=dev); db2 = torch.empty(mid, dtype=torch.float32, device=dev)
            dS2 = torch.empty((groups, 2, mid), dtype=torch.float64, device=dev)
            call("sdhip_bn_finalize_bwd", ptr(dsc2), ptr(dsh2), ptr(layer.norm2.weight), ptr(mu2), ptr(iv2), ptr(dg2), ptr(db2),
                 ptr(dS2), mid, 0, mid, groups, float(count), int(training), st)
            call("sdhip_stats_fix", ptr(gp2), mid, ptr(y1), mid, ptr(gp2), mid, ptr(dS2), mid, npix, mid, groups, dt, st)
            # (6) conv1 (1x1): data gradient w.r.t. relu(norm1(slab[:Cin])), weight gradient
            wd1 = ops.packed_weight(layer.conv1.weight, 'conv', 'dgrad', dtype)
            gp1 = ops.empty_nhwc(B, Cin, H, W, dtype, dev)
            ops._conv_launch(gp2, mid, wd1, gp1, Cin, None, None, None, None, B, H, W, mid, H, W, Cin, 1, 1, 1, 1, 0, 0,
                             False, 1, 0, False)
            gw1 = _wgrad(slab, Ct, gp2, mid, layer.conv1.weight, B, H, W, Cin, mid, 1, 0, sc1, sh1, groups, dt)
            # (7) through relu + norm1's affine, accumulated into the slab gradient; (8) norm1's statistics
            dsc1 = torch.empty_like(sc1); dsh1 = torch.empty_like(sh1)
            call("sdhip_affine_act_bwd", ptr(gp1), Cin, ptr(slab), Ct, ptr(g_slab), Ct, ptr(sc1), ptr(sh1), ptr(dsc1), ptr(dsh1),
                 npix, Cin, groups, 1, 1, dt, st)
            dg1 = torch.empty(Cin, dtype=torch.float32, device=dev); db1 = torch.empty(Cin, dtype=torch.float32, device=dev)
            call("sdhip_bn_finalize_bwd", ptr(dsc1), ptr(dsh1), ptr(layer.norm1.weight), ptr(mu1), ptr(iv1), ptr(dg1), ptr(db1),
                 ptr(dS), Ct, 1, Cin, groups, float(count), int(training), st)
            grads.append((dg1, db1, gw1, dg2, db2, gw2))
        gx0 = ops.empty_nhwc(B, C0, H, W, dtype, dev)
        call("sdhip_stats_fix", ptr(g_slab), Ct, ptr(slab), Ct, ptr(gx0), C0, ptr(dS), Ct, npix, C0, groups, dt, st)
        flat = []
        for g6 in reversed(grads):
            flat.extend(g6)
        return (gx0, None, None) + tuple(flat)


class _DenseLayer(nn.Module):
    """models/densenet.py:25-93 — parameter container; the block drives the kernels."""

    def __init__(self, cin, growth, bn_size):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(cin)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(cin, bn_size * growth, 1, bias=False)
        self.norm2 = nn.BatchNorm2d(bn_size * growth)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(bn_size * growth, growth, 3, padding=1, bias=False)


class _DenseBlock(nn.ModuleDict):
    """models/densenet.py:96-116."""

    def __init__(self, n, cin, bn_size, growth):
        super().__init__()
        for i in range(n):
            self['denselayer%d' % (i + 1)] = _DenseLayer(cin + i * growth, growth, bn_size)

    def forward(self, x, groups=1):
        params = []
        for layer in self.values():
            params += [layer.norm1.weight, layer.norm1.bias, layer.conv1.weight,
                       layer.norm2.weight, layer.norm2.bias, layer.conv2.weight]
        return _DenseBlockFn.apply(x, self, groups, *params)   # (slab, statistics)


class _Transition(nn.Sequential):
    """models/densenet.py:119-128: norm + relu fused into the 1x1 conv's prologue (pool stays outside)."""

    def __init__(self, cin, cout):
        super().__init__(OrderedDict(norm=nn.BatchNorm2d(cin), relu=nn.ReLU(inplace=True),
                                     conv=nn.Conv2d(cin, cout, 1, bias=False)))

    def forward(self, slab, stats, groups=1):
        B, C, H, W = slab.shape
        scale, shift = ops.bn_scale_shift(self.norm, stats if self.norm.training else None, (B // groups) * H * W, groups)
        return ops.conv2d(slab, self.conv.weight, None, kind='conv', padding=0, in_scale=scale, in_shift=shift,
                          in_relu=True, groups=groups)


class DenseNet(nn.Module):
    """models/densenet.py:131-245: returns the five taps; tap 0 is the raw conv0 output, taps 1-3 the transition
    outputs before the 2x2 average pool, tap 4 relu(norm5(.))."""

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
        self.classifier = nn.Linear(c, num_classes)   # never called by the reference either (kept for the state_dict)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.constant_(m.bias, 0)

    def forward(self, x, groups=1):
        from .nn import bn_apply
        n0 = self.features.norm0
        if n0.training:
            c0, st = ops.conv2d(x, self.conv0.weight, None, kind='conv', stride=2, padding=3, want_stats=True, groups=groups)
        else:
            c0, st = ops.conv2d(x, self.conv0.weight, None, kind='conv', stride=2, padding=3, groups=groups), None
        taps = [c0]
        f = ops.maxpool3s2(bn_apply(n0, c0, st, act=1, groups=groups))
        stats = None
        for i, blk in enumerate(self.denseblock):
            if i % 2 == 0:
                f, stats = blk(f, groups)
            else:
                f = blk(f, stats, groups)
                taps.append(f)
                f = ops.avgpool(f, 2)
        taps.append(bn_apply(self.norm5, f, stats, act=1, groups=groups))
        return taps


def densenet121(pretrained=False, progress=True, **kwargs):
    if pretrained:
        raise RuntimeError("pretrained=True downloads ImageNet weights (models/densenet.py:256); load a state_dict instead")
    return DenseNet(32, (6, 12, 24, 16), 64, **kwargs)
