"""Autograd bindings of the libsdhip.so kernels.

Tensors keep the reference's logical NCHW shapes but live in channels-last
(NHWC) memory, possibly as a channel slice of a wider slab; `nhwc_view`
returns the pixel stride the C ABI wants.  Nothing in this file computes on the
CPU or through ATen kernels: a non-GPU tensor is an error.
"""
import torch

from . import _lib
from ._lib import call, dtype_code, ptr, stream_ptr


def _require_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.SdhipError("sdhip ops need GPU tensors (got %s); there is no CPU path" % t.device)


def nhwc_view(x):
    """Return (tensor, ld): x itself if its memory is NHWC with a uniform pixel stride, else a channels-last copy."""
    B, C, H, W = x.shape
    sb, sc, sh, sw = x.stride()
    ld = sw
    ok = (sc == 1 or C == 1) and ld >= C and (sh == W * ld or H == 1) and (sb == H * W * ld or B == 1)
    if W == 1:
        ok = False if (sc != 1 and C != 1) else (sh >= C and (sb == H * sh or B == 1))
        ld = sh if ok else ld
    if not ok:
        x = x.contiguous(memory_format=torch.channels_last)
        if x.stride(1) != 1 and C != 1:  # degenerate shapes: force an explicit NHWC buffer
            x = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        ld = C
    return x, ld


def empty_nhwc(B, C, H, W, dtype, device):
    return torch.empty((B, H, W, C), dtype=dtype, device=device).permute(0, 3, 1, 2)


class _CorrFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, in1, in2, PH, PW, dil):
        _require_gpu(in1, in2)
        if in1.shape != in2.shape or in1.dtype != in2.dtype:
            raise _lib.SdhipError("correlation inputs must have the same shape and dtype")
        B, C, H, W = in1.shape
        a, lda = nhwc_view(in1)
        b, ldb = nhwc_view(in2)
        if lda != ldb:
            a = a.contiguous(memory_format=torch.channels_last); b = b.contiguous(memory_format=torch.channels_last)
            lda = ldb = C
        out = torch.empty((B, H, W, PH, PW), dtype=in1.dtype, device=in1.device)
        call("sdhip_corr_fwd", ptr(a), ptr(b), ptr(out), B, H, W, C, lda, PH, PW, dil, PH * PW,
             dtype_code(in1), stream_ptr())
        ctx.save_for_backward(a, b)
        ctx.cfg = (PH, PW, dil, lda)
        # (B,PH,PW,H,W) view of NHWC memory, displacement fastest
        return out.permute(0, 3, 4, 1, 2)

    @staticmethod
    def backward(ctx, gout):
        a, b = ctx.saved_tensors
        PH, PW, dil, ld = ctx.cfg
        B, C, H, W = a.shape
        g = gout.permute(0, 3, 4, 1, 2)  # (B,H,W,PH,PW)
        if not g.is_contiguous():
            g = g.contiguous()
        if ld != C:
            a = a.contiguous(memory_format=torch.channels_last); b = b.contiguous(memory_format=torch.channels_last)
            ld = C
        ga = empty_nhwc(B, C, H, W, a.dtype, a.device)
        gb = empty_nhwc(B, C, H, W, a.dtype, a.device)
        call("sdhip_corr_bwd", ptr(a), ptr(b), ptr(g), ptr(ga), ptr(gb), B, H, W, C, ld, PH, PW, dil,
             PH * PW, dtype_code(a), stream_ptr())
        return ga, gb, None, None, None


def correlation(in1, in2, patch_h, patch_w, dilation_patch=1):
    return _CorrFn.apply(in1, in2, patch_h, patch_w, dilation_patch)
