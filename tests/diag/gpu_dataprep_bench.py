"""Time sdhip_prepare_sample (HIP events over 50 launches, inputs resident in HBM) and the CPU path it replaces
(the numpy oracle restatement of the reference loader's post-decode work) on the same sample."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pmt_learning_for_semantic_segmentation_and_disparity_amd import _lib  # noqa: E402
from pmt_learning_for_semantic_segmentation_and_disparity_amd.data import SamplePreparer  # noqa: E402
from oracle import data_ref as DR  # noqa: E402

rng = np.random.default_rng(0)
res = []
for (H, W, oh, ow, dt) in [(1024, 2048, 512, 1024, torch.bfloat16), (1024, 2048, 1024, 2048, torch.float32), (720, 1280, 256, 512, torch.bfloat16)]:
    left = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    right = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    seg = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    depth = rng.uniform(-1, 30, (H, W)).astype(np.float32)
    pfm = b"Pf\n%d %d\n-1.0\n" % (W, H) + np.flipud(depth).tobytes()
    norm = np.array([[0, 0, 0], [1, 1, 1]], dtype=np.float32)
    crop = ((H - oh) // 2, (W - ow) // 2, oh, ow)
    sp = SamplePreparer("roses", 2, 192, "linear", norm, dtype=dt, device="cuda:0")
    batch = sp.alloc_batch(1, oh, ow)
    # stage once, then re-launch the kernel on the resident buffers (the H2D copy is PCIe, not the kernel)
    calls = []
    orig = _lib.call
    def spy(name, *a):
        calls.append((name, a))
        return orig(name, *a)
    import pmt_learning_for_semantic_segmentation_and_disparity_amd.data as D
    D.call = spy
    t0 = time.perf_counter()
    sp.prepare_into(batch, 0, left, right, seg, pfm, crop)
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t0
    D.call = orig
    name, args = calls[0]
    for _ in range(5):
        orig(name, *args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        orig(name, *args)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    es = 2 if dt == torch.bfloat16 else 4
    nbytes = oh * ow * (3 + 3 + 1 + 4 + 2 * 3 * es + 2 * 4 + 4)     # bytes the algorithm needs: 1 seg byte (3 fetched as one sector)
    t0 = time.perf_counter()
    DR.prepare_sample(left, right, seg, pfm, "roses", 2, 192.0, "linear", norm, crop)
    t_cpu = time.perf_counter() - t0
    res.append(dict(src=[H, W], crop=[oh, ow], dtype=str(dt), kernel_us=us, alg_bytes=nbytes, GBps=nbytes / us / 1e3,
                    frac_hbm=nbytes / us / 1e3 / 8000.0, first_call_incl_h2d_ms=t_first * 1e3, cpu_numpy_ms=t_cpu * 1e3))
    print(json.dumps(res[-1]))
os.makedirs(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "dataprep_bench.json"), "w"), indent=1)
