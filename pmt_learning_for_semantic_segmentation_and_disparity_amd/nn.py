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
