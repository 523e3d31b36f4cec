"""Time the correlation kernels (forward, backward) at the step's shapes: tiled MFMA form vs the wave-per-pixel form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops, _lib

def bench(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (B, C, H, W, PH, PW) in ((8, 352, 32, 64, 17, 17), (8, 448, 32, 64, 17, 17), (8, 352, 32, 64, 1, 17), (8, 256, 16, 32, 1, 17), (4, 352, 64, 120, 1, 17)):
    a = torch.randn(B, H, W, C, device="cuda").bfloat16().permute(0, 3, 1, 2).requires_grad_(True)
    b = torch.randn(B, H, W, C, device="cuda").bfloat16().permute(0, 3, 1, 2).requires_grad_(True)
    for tiled in (True, False):
        if tiled:
            os.environ.pop("SDHIP_CORR_NO_TILED", None)
        else:
            os.environ["SDHIP_CORR_NO_TILED"] = "1"
        _lib.reload_diag()
        y = ops.correlation(a, b, PH, PW, 1)
        g = torch.randn_like(y)
        tf = bench(lambda: ops.correlation(a, b, PH, PW, 1))
        tb = bench(lambda: torch.autograd.grad(y, (a, b), g, retain_graph=True))
        print("B=%d C=%d %dx%d patch (%d,%d) %s: fwd %.1f us  bwd %.1f us" % (B, C, H, W, PH, PW, "tiled" if tiled else "wave/pixel", tf, tb), flush=True)
os.environ.pop("SDHIP_CORR_NO_TILED", None)
