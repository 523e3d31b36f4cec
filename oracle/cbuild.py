"""ORACLE helper (test infrastructure only): build + load the plain-C restatements with gcc."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "liboracle_ref.so")
SRCS = [os.path.join(HERE, "corr_ref.c")]


def build(force=False):
    os.makedirs(OUT, exist_ok=True)
    stale = force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in SRCS)
    if stale:
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", LIB] + SRCS)
    return LIB


def load():
    lib = ctypes.CDLL(build())
    fp, i = ctypes.POINTER(ctypes.c_float), ctypes.c_int
    lib.corr_ref_forward.argtypes = [fp, fp, fp] + [i] * 16
    lib.corr_ref_backward.argtypes = [fp, fp, fp, fp, fp] + [i] * 16
    lib.corr_ref_forward.restype = None
    lib.corr_ref_backward.restype = None
    return lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def corr_forward(lib, in1, in2, patch, dil_patch=1):
    """numpy NCHW float32 -> (B,PH,PW,H,W); kernel_size=1, stride=1, padding=0."""
    import numpy as np
    B, C, H, W = in1.shape
    PH, PW = patch
    out = np.empty((B, PH, PW, H, W), np.float32)
    lib.corr_ref_forward(_fp(np.ascontiguousarray(in1)), _fp(np.ascontiguousarray(in2)), _fp(out), B, C, H, W,
                         1, 1, PH, PW, 0, 0, 1, 1, dil_patch, dil_patch, 1, 1)
    return out


def corr_backward(lib, in1, in2, gout, patch, dil_patch=1):
    import numpy as np
    B, C, H, W = in1.shape
    PH, PW = patch
    g1, g2 = np.empty_like(in1), np.empty_like(in2)
    lib.corr_ref_backward(_fp(np.ascontiguousarray(in1)), _fp(np.ascontiguousarray(in2)), _fp(np.ascontiguousarray(gout)),
                          _fp(g1), _fp(g2), B, C, H, W, 1, 1, PH, PW, 0, 0, 1, 1, dil_patch, dil_patch, 1, 1)
    return g1, g2
