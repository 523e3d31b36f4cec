"""ORACLE helper (test infrastructure only): deterministic, version-independent
parameter / input generators, so golden fixtures need not carry 47 MB of weights.

Values depend only on (seed, state_dict key, shape) through numpy's MT19937
RandomState, which is stable across numpy versions.
"""
import zlib

import numpy as np
import torch


def _rs(seed, key):
    return np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def fill_state_dict(module, seed=0):
    """Overwrite every floating entry of module.state_dict() with deterministic values."""
    sd = module.state_dict()
    new = {}
    for k in sorted(sd.keys()):
        v = sd[k]
        if not torch.is_floating_point(v):
            new[k] = v.clone()
            continue
        rs = _rs(seed, k)
        shape = tuple(v.shape)
        leaf = k.rsplit('.', 1)[-1]
        if leaf == 'running_var':
            a = rs.uniform(0.5, 1.5, size=shape)
        elif leaf == 'running_mean':
            a = rs.normal(0.0, 0.1, size=shape)
        elif v.dim() == 1 and leaf == 'weight':      # BN gamma
            a = rs.uniform(0.6, 1.4, size=shape)
        elif v.dim() == 1:                           # biases / BN beta
            a = rs.normal(0.0, 0.1, size=shape)
        else:                                        # conv / linear weights: He-like, fan-in scaled
            fan_in = int(np.prod(shape[1:])) if v.dim() > 1 else shape[0]
            if 'ct2d' in k or 'ConvTranspose' in k:  # (Cin, Cout, kh, kw)
                fan_in = shape[0] * int(np.prod(shape[2:]))
            a = rs.normal(0.0, np.sqrt(2.0 / max(fan_in, 1)), size=shape)
        new[k] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).reshape(shape)
    module.load_state_dict(new)
    return module


def rand_input(seed, name, shape, lo=0.0, hi=1.0):
    rs = _rs(seed, 'input:' + name)
    return torch.from_numpy(rs.uniform(lo, hi, size=shape).astype(np.float32))


def randn_input(seed, name, shape, std=1.0):
    rs = _rs(seed, 'input:' + name)
    return torch.from_numpy((rs.normal(0.0, std, size=shape)).astype(np.float32))
