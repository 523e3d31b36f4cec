"""1x1 weight-gradient shapes of the DenseNet towers (GEMM-like: few pixels, many channels), timed one by one."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
dtype = torch.bfloat16
shapes = [(16, 16, 32, 1024, 128, 1), (16, 16, 32, 1024, 512, 1), (16, 32, 64, 512, 128, 1), (16, 32, 64, 512, 256, 1),
          (16, 64, 128, 256, 128, 1), (16, 64, 128, 192, 128, 1), (16, 64, 128, 128, 128, 0), (8, 64, 128, 512, 128, 0),
          (16, 8, 16, 1024, 128, 1), (16, 128, 256, 64, 128, 0)]
for (B, H, W, Cin, Cout, pro) in shapes:
    x = torch.randn(B, H, W, Cin, device="cuda").to(dtype).permute(0, 3, 1, 2)
    g = (torch.randn(B, H, W, Cout, device="cuda") * 0.1).to(dtype).permute(0, 3, 1, 2)
    w = torch.zeros(Cout, Cin, 1, 1, device="cuda")
    sc = (torch.rand(1, Cin, device="cuda") + 0.5) if pro else None
    sh = (torch.rand(1, Cin, device="cuda") - 0.5) if pro else None
    spec = ops.ConvSpec('conv', 1, 1, 1, 1, 0, 0, H, W)
    for _ in range(3):
        ops._wgrad_impl(x, Cin, g, Cout, w, None, spec, sc, sh, bool(pro), 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops._wgrad_impl(x, Cin, g, Cout, w, None, spec, sc, sh, bool(pro), 1)
    e1.record()
    torch.cuda.synchronize()
    npix = B * H * W
    ideal = max(2.0 * npix * Cin * Cout / 2.5e15, (npix * (Cin + Cout) * 2.0) / 8e12) * 1e6
    print("B%d %dx%d %4d->%3d pro%d: %.1f us per call (all launches), ideal %.1f us" % (B, H, W, Cin, Cout, pro, e0.elapsed_time(e1) * 50, ideal), flush=True)
