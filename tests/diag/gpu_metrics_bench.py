"""Time sdhip_step_metrics at BASELINE config 2 size (B=8, 256x512, bf16 outputs) with HIP events, and the host-side
block it replaces (D2H copies + the numpy oracle restatement of the reference's functions) beside it."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pmt_learning_for_semantic_segmentation_and_disparity_amd.metrics import StepMetrics  # noqa: E402

dev = torch.device("cuda:0")
res = []
for (B, L, Ct, H, W, dt) in [(8, 2, 2, 256, 512, torch.bfloat16), (8, 2, 2, 256, 512, torch.float32), (4, 19, 20, 512, 1024, torch.bfloat16)]:
    g = torch.Generator(device="cuda").manual_seed(0)
    logits = torch.randn(B, L, H, W, device=dev, generator=g).to(dt).contiguous(memory_format=torch.channels_last)
    cls = torch.randint(0, Ct, (B, H, W), device=dev, generator=g)
    seg_full = torch.nn.functional.one_hot(cls, Ct).permute(0, 3, 1, 2).float()      # NHWC memory, NCHW shape
    disp = torch.rand(B, 1, H, W, device=dev, generator=g) * 8 + 0.1
    dp = (disp + torch.randn(B, 1, H, W, device=dev, generator=g)).to(dt)
    m = StepMetrics(L, max_disp=1.0, device=dev)
    for _ in range(5):
        m.update(logits, seg_full, dp, disp)
    torch.cuda.synchronize()
    n = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        m.update(logits, seg_full, dp, disp)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    es = logits.element_size()
    nbytes = B * H * W * (L * es + Ct * 4 + es + 4)
    # the host block of losses/multiLosses.py:116-125,146-154: 5 D2H copies + numpy
    from oracle import metrics_ref as MR
    t0 = time.perf_counter()
    a = [t.float().cpu().numpy() for t in (logits, torch.log_softmax(logits.float(), 1), seg_full, dp, disp)]
    t1 = time.perf_counter()
    MR.step_metrics(a[0], a[2], a[3], a[4], L, 1.0)
    t2 = time.perf_counter()
    res.append(dict(B=B, L=L, H=H, W=W, dtype=str(dt), kernel_us=us, bytes=nbytes, GBps=nbytes / us / 1e3,
                    frac_hbm=nbytes / us / 1e3 / 8000.0, host_copy_ms=(t1 - t0) * 1e3, host_numpy_ms=(t2 - t1) * 1e3))
    print(json.dumps(res[-1]))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/metrics_bench.json", "w"), indent=1)
