"""ctypes binding of libsdhip.so — the only way the package reaches the GPU.

There is no CPU or eager-PyTorch fallback: if the shared library is missing or
fails to load, importing this module raises.  `import torch` happens first on
purpose: libsdhip.so needs `libamdhip64.so.7` by soname and must bind to the HIP
runtime PyTorch-ROCm has already mapped, so that stream handles and device
pointers are shared between the two.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the dlopen below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsdhip.so")

F32, BF16 = 0, 1
ERR_ARG, ERR_LAUNCH, ERR_UNSUPPORTED = -1, -2, -3   # include/sdhip.h
NREP = int(os.environ.get("SDHIP_TUNE_NREP", "32"))   # statistics replicas the kernels spread their atomics over (env: tuning only)


class SdhipError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libsdhip.so not found at %s — run `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc, gfx950). This package has no fallback path." % LIB_PATH)

_lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)

_p, _i, _f, _d, _l = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_long
_lib.sdhip_last_error.restype = ctypes.c_char_p
_lib.sdhip_abi_version.restype = _i

# name -> argtypes; every entry of include/sdhip.h is listed here (tests/test_abi.py checks).
SIGNATURES = {
    "sdhip_corr_fwd": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "sdhip_corr_bwd": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "sdhip_conv_pack_weights": [_p, _p, _i, _i, _i, _l, _l, _i, _i, _p],
    "sdhip_conv_unpack_wgrad": [_p, _p, _i, _i, _i, _l, _l, _i, _i, _i, _p],
    "sdhip_conv2d_fwd": [_p, _p, _p, _p, _p, _p, _p] + [_i] * 27 + [_p],
    "sdhip_conv2d_fwd_phase": [_p, _p, _p, _p, _i, _i] + [_i] * 16 + [_p],
    "sdhip_conv2d_wgrad": [_p, _p, _p, _p, _p, _p] + [_i] * 24 + [_p],
    "sdhip_conv2d_wgrad_group": [_p, _i, _i, _i, _p],
    "sdhip_conv1x1_cat_fwd": [_p, _i, _i, _i, _p, _i, _i, _i, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p],
    "sdhip_conv_pack_batch": [_p, _i, _i, _p],
    "sdhip_conv_unpack_batch": [_p, _i, _i, _p],
    "sdhip_rowpool_max_fwd": [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p],
    "sdhip_rowpool_max_bwd": [_p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "sdhip_mul_rows_fwd": [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p],
    "sdhip_mul_rows_bwd": [_p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p],
    "sdhip_dropout_channels": [_p, _i, _p, _i, _p, _l, _i, _i, _i, _f, _i, _p],
    "sdhip_channel_stats": [_p, _i, _p, _i, _i, _l, _i, _i, _i, _i, _p],
    "sdhip_stats_replica_sum": [_p, _p, _i, _i, _i, _i, _i, _p],
    "sdhip_bn_fold_finalize": [_p, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _d, _f, _f, _p],
    "sdhip_bn_finalize": [_p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _d, _f, _f, _p],
    "sdhip_stats_fix_fin": [_p, _i, _p, _i, _p, _i, _l, _p, _i, _i, _p, _p, _i, _p, _p, _p, _p, _p, _i, _f, _i, _i, _d, _i, _p],
    "sdhip_bn_finalize_bwd": [_p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _d, _i, _p],
    "sdhip_affine_act": [_p, _i, _p, _i, _p, _i, _p, _p, _l, _i, _i, _i, _i, _p],
    "sdhip_affine_act_bwd": [_p, _i, _p, _i, _p, _i, _p, _p, _p, _p, _i, _l, _i, _i, _i, _i, _i, _i, _p],
    "sdhip_maxpool3s2_fwd": [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _p],
    "sdhip_maxpool3s2_bwd": [_p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "sdhip_avgpool_fwd": [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "sdhip_avgpool_bwd": [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "sdhip_resize_fwd": [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _i, _p],
    "sdhip_resize_bwd": [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _f, _f, _i, _p],
    "sdhip_mul_bcast_fwd": [_p, _i, _p, _i, _p, _i, _l, _i, _i, _p],
    "sdhip_mul_bcast_bwd": [_p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _l, _i, _i, _p],
    "sdhip_adam_step": [_p, _p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _f, _p],
    "sdhip_ce_loss": [_p, _i, _p, _i, _p, _i, _p, _l, _i, _f, _i, _p],
    "sdhip_dropout": [_p, _p, _p, _l, _l, _f, _i, _p],
    "sdhip_lovasz_softmax": [_p, _i, _p, _i, _p, _i, _p, _l, _i, _f, _p, _l, _i, _i, _p],
    "sdhip_stuff": [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "sdhip_cost_volume_fwd": [_p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p],
    "sdhip_cost_volume_bwd": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "sdhip_softargmin_fwd": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "sdhip_softargmin_bwd": [_p, _p, _p, _p, _p, _l, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "sdhip_log_softmax_fwd": [_p, _i, _p, _i, _l, _i, _i, _p],
    "sdhip_log_softmax_bwd": [_p, _i, _p, _i, _p, _i, _l, _i, _i, _p],
    "sdhip_l1_loss": [_p, _p, _p, _p, _l, _f, _i, _i, _p],
    "sdhip_stats_fix": [_p, _i, _p, _i, _p, _i, _p, _i, _l, _i, _i, _i, _p],
    "sdhip_bn_bwd_apply": [_p, _i, _p, _i, _p, _i, _p, _p, _p, _i, _l, _i, _i, _i, _i, _p],
    "sdhip_affine_act_bn": [_p, _i, _p, _i, _p, _i, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _l, _i, _i, _d, _f, _f, _i, _i, _p],
    "sdhip_step_metrics": [_p, _i, _p, _i, _i, _p, _p, _p, _p, _i, _i, _i, _l, _i, _f, _i, _i, _p],
    "sdhip_double_left_sample": [_p, _p, _i, _p, _i, _i, _p, _i, _i, _i, _p],
    "sdhip_prepare_sample": [_p, _p, _l, _i, _p, _l, _i, _i, _i, _i, _p, _p, _l, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _i, _p, _p,
                             _p, _p, _i, _p, _i, _i, _p, _i, _p],
    "sdhip_flip_sample": [_p, _p, _i, _p, _i, _i, _p, _i, _i, _p, _l, _i, _p],
    "sdhip_bn_bwd_apply_fin": [_p, _i, _p, _i, _p, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _i, _f, _l, _i, _i, _d, _i, _i, _p],
    "sdhip_bn_bwd_apply_fin_d": [_p, _i, _p, _i, _p, _i, _p, _p, _p, _i, _p, _p, _p, _p, _p, _i, _f, _l, _i, _i, _d, _i, _i, _p],
    "sdhip_conv2d_fwd_bnpro": [_p, _p, _p, _p, _i, _i, _p, _i, _i, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _d, _f, _f] + [_i] * 15 + [_p],
    "sdhip_conv2d_fwd_add": [_p, _p, _p, _p, _i] + [_i] * 14 + [_p],
    "sdhip_conv2d_fwd_bnbwd": [_p, _p, _p, _p, _i, _i, _p, _i, _p, _p, _p, _i] + [_i] * 17 + [_p],
}
_lib.sdhip_lovasz_workspace_bytes.argtypes = [_l, _i]
_lib.sdhip_lovasz_workspace_bytes.restype = _l
_lib.sdhip_flip_sample_workspace_bytes.argtypes = [_i, _i, _i]
_lib.sdhip_flip_sample_workspace_bytes.restype = _l
_lib.sdhip_softargmin_bwd_workspace_floats.argtypes = [_i] * 7
_lib.sdhip_softargmin_bwd_workspace_floats.restype = _l
_lib.sdhip_conv_packed_elems.argtypes = [_i, _i, _i, _i]
_lib.sdhip_conv_packed_elems.restype = _l
for _name, _args in SIGNATURES.items():
    _fn = getattr(_lib, _name)
    _fn.argtypes = _args
    _fn.restype = _i


class WgradItem(ctypes.Structure):
    """SdhipWgradItem of include/sdhip.h."""
    _fields_ = [(n, _p) for n in ("x", "dy", "dw_packed", "dbias", "in_scale", "in_shift")] + \
               [(n, _i) for n in ("B", "H", "W", "Cin", "ldx", "Ho", "Wo", "Cout", "lddy", "kh", "kw", "stride", "dil", "pad_t", "pad_l",
                                  "D", "Do", "kd", "sd", "pad_d", "in_relu", "groups")]


def lovasz_workspace_bytes(npix, C):
    return _lib.sdhip_lovasz_workspace_bytes(npix, C)


def packed_elems(M, K, T, dt):
    return _lib.sdhip_conv_packed_elems(M, K, T, dt)


_lib.sdhip_diag_reload.restype = None
_lib.sdhip_abort_capture.argtypes = [_p]
_lib.sdhip_abort_capture.restype = _i


_lib.sdhip_graph_node_counts.argtypes = [_p, ctypes.POINTER(_i)]
_lib.sdhip_graph_node_counts.restype = _i


def graph_node_counts(graph):
    """{kernel, memset, memcpy, other, total} of a torch.cuda.CUDAGraph created with keep_graph=True."""
    c = (_i * 4)()
    n = _lib.sdhip_graph_node_counts(ctypes.c_void_p(graph.raw_cuda_graph()), c)
    if n < 0:
        raise SdhipError("sdhip_graph_node_counts failed: %s" % _lib.sdhip_last_error().decode())
    return {"kernel": c[0], "memset": c[1], "memcpy": c[2], "other": c[3], "total": n}


def abort_capture(stream):
    """End a broken hipGraph capture on `stream` (a torch.cuda.Stream); returns 1 if one was open (see include/sdhip.h)."""
    rc = _lib.sdhip_abort_capture(ctypes.c_void_p(stream.cuda_stream))
    if rc < 0:
        raise SdhipError("sdhip_abort_capture failed: %s" % _lib.sdhip_last_error().decode())
    return rc


def reload_diag():
    """Re-read the SDHIP_* diagnostic environment switches (they are otherwise fixed at library load)."""
    _lib.sdhip_diag_reload()


def _diag_switch(name):
    """Python-side diagnostic switches are read once, at import, and announced."""
    v = os.environ.get(name)
    if v:
        import sys
        sys.stderr.write("[sdhip] diagnostic switch %s=%s is set: this is not the production path\n" % (name, v))
    return v


DIAG_NO_FUSED_BN = bool(_diag_switch("SDHIP_DIAG_NO_FUSED_BN"))
DIAG_NO_SIDE = bool(_diag_switch("SDHIP_DIAG_NO_SIDE"))
DIAG_NO_GRAD_SLOTS = bool(_diag_switch("SDHIP_DIAG_NO_GRAD_SLOTS"))
DIAG_NO_WGRAD_GROUP = bool(_diag_switch("SDHIP_DIAG_NO_WGRAD_GROUP"))
DIAG_NO_PHASE_DGRAD = bool(_diag_switch("SDHIP_DIAG_NO_PHASE_DGRAD"))     # data gradient of stride-2 3-D convolutions over the zero-stuffed dY (A/B)
TUNE_WGRAD_OVERLAP = bool(_diag_switch("SDHIP_TUNE_WGRAD_OVERLAP"))     # measured and not kept as a default: see train.TrainStep
TUNE_OVERLAP_WG = int(_diag_switch("SDHIP_TUNE_OVERLAP_WG") or 128)
DIAG_NO_BN_SLOTS = bool(_diag_switch("SDHIP_DIAG_NO_BN_SLOTS"))
TUNE_FUSE1_MAX_PIX = int(_diag_switch("SDHIP_TUNE_FUSE1_MAX_PIX") or 32768)
TUNE_PRO_MAX_PIX = int(_diag_switch("SDHIP_TUNE_PRO_MAX_PIX") or 32768)
TUNE_BN_SLOT_MAX_MB = int(_diag_switch("SDHIP_TUNE_BN_SLOT_MAX_MB") or 40)
DIAG_NO_BNPRO = bool(_diag_switch("SDHIP_DIAG_NO_BNPRO"))
DIAG_NO_BNBWD_EPILOGUE = bool(_diag_switch("SDHIP_DIAG_NO_BNBWD_EPILOGUE"))
DIAG_STEM_S2D = _diag_switch("SDHIP_STEM_S2D")


def abi_version():
    return _lib.sdhip_abi_version()


def call(name, *args):
    """Invoke a C entry point; raise SdhipError with the library's message on failure."""
    rc = getattr(_lib, name)(*args)
    if rc != 0:
        raise SdhipError("%s failed (%d): %s" % (name, rc, _lib.sdhip_last_error().decode()))


def dtype_code(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise SdhipError("unsupported dtype %s (f32 and bf16 only)" % t.dtype)


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
