"""Drop-in nn.Module surface of the hot path (same class names, constructor
signatures and state_dict keys as the reference), every forward running on the
hand-written HIP kernels of libsdhip.so.
"""
import torch
import torch.nn as nn

from . import ops


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


def _no_dropout(p, training):
    if p and training:
        raise NotImplementedError("Dropout(p>0) in training mode is not on the native path yet (the shipped recipe uses p=0)")


def bn_apply(bn, y, stats, act=0, residual=None, groups=1):
    """BatchNorm2d (+activation, + skip add) of a raw conv output whose batch statistics rode on the conv epilogue."""
    B, C, H, W = y.shape
    scale, shift = ops.bn_scale_shift(bn, stats, (B // groups) * H * W, groups)
    return ops.affine_act(y, scale, shift, residual, act, groups)


class conv2dSame(nn.Module):
    """models/torch_model.py:236-281. `c2d` is kept as the parameter container (state_dict key `c2d.weight`)."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding='valid', dilation=1, bias=False):
        super().__init__()
        self.padding = padding
        self.c2d = nn.Conv2d(in_channel, out_channel, kernel_size, stride=stride, dilation=dilation, bias=bias)
        _he_init([self.c2d])

    def run(self, x, act=0, want_stats=False, groups=1, in_scale=None, in_shift=None, in_relu=False):
        c = self.c2d
        return ops.conv2d(x, c.weight, c.bias, kind='conv', stride=c.stride[0], dilation=c.dilation[0],
                          padding='same' if self.padding == 'same' else 0, act=act, want_stats=want_stats, groups=groups,
                          in_scale=in_scale, in_shift=in_shift, in_relu=in_relu)

    def forward(self, x):
        return self.run(x)


class ConvTranspose2dSame(nn.Module):
    """models/torch_model.py:284-349 (stride 1: a correlation with flipped, transposed weights — no (k-1) border
    is computed and thrown away, no crop copy)."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding='valid', dilation=1, bias=False, init_he=True):
        super().__init__()
        self.padding = padding
        self.ct2d = nn.ConvTranspose2d(in_channel, out_channel, kernel_size, stride=stride, dilation=dilation, bias=bias)

    def run(self, x, act=0, want_stats=False, groups=1):
        c = self.ct2d
        if self.padding != 'same' or c.stride[0] != 1:
            raise NotImplementedError("ConvTranspose2dSame: only padding='same', stride=1 is on the native path")
        return ops.conv2d(x, c.weight, c.bias, kind='deconv', stride=1, dilation=c.dilation[0], padding='ctsame',
                          act=act, want_stats=want_stats, groups=groups)

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

    def fused(self, x, act=0, residual=None, groups=1):
        """conv -> BatchNorm (batch statistics from the conv epilogue) -> activation (-> + residual), 2 launches."""
        conv = self.layers[0]
        if len(self.layers) == 1:
            y = conv.run(x, act=act, groups=groups)
            return y if residual is None else ops.affine_act(y, None, None, residual, 0, 1)
        bn = self.layers[1]
        if bn.training:
            y, stats = conv.run(x, want_stats=True, groups=groups)
        else:
            y, stats = conv.run(x, groups=groups), None
        return bn_apply(bn, y, stats, act, residual, groups)

    def forward(self, x):
        return self.fused(x)


class deconvbn(convbn):
    """models/dsnet_t2.py:48-77."""
    _conv = ConvTranspose2dSame


def _act_block(block, p=0.0):
    return nn.Sequential(block, nn.ReLU(inplace=True), nn.Dropout(p=p))


def run_act_block(seq, x, residual=None, groups=1):
    """Sequential(convbn|deconvbn, ReLU[, Dropout]) as one fused conv+BN+ReLU(+skip)."""
    if len(seq) > 2:
        _no_dropout(seq[2].p, seq.training)
    return seq[0].fused(x, act=1, residual=residual, groups=groups)


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
        x1 = run_act_block(self.c1, x, groups=groups)
        x2 = run_act_block(self.c2, x1, groups=groups)
        x = run_act_block(self.c3, x2, groups=groups)
        x = run_act_block(self.d3, x, residual=x2, groups=groups)
        x = run_act_block(self.d4, x, residual=x1, groups=groups)
        return run_act_block(self.d5, x, groups=groups) if self.lastLayer else x
