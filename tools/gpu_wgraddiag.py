"""Weight-gradient kernels: general (SDHIP_WGRAD_GENERIC=1) vs bf16 fast path — parity + time per shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
dtype = torch.bfloat16
# B H W Cin Cout k dil prologue
SHAPES = [(8, 256, 512, 64, 64, 5, 1, 0), (8, 256, 512, 32, 32, 3, 1, 0), (8, 256, 512, 64, 64, 3, 1, 0), (8, 64, 128, 64, 64, 3, 1, 0),
          (16, 64, 128, 128, 32, 3, 1, 1), (16, 32, 64, 128, 32, 3, 1, 1), (16, 16, 32, 128, 32, 3, 1, 1), (16, 8, 16, 128, 32, 3, 1, 1),
          (16, 32, 64, 512, 256, 1, 1, 1), (16, 16, 32, 1024, 512, 1, 1, 1), (16, 64, 128, 256, 128, 1, 1, 1), (8, 64, 128, 512, 128, 1, 1, 0),
          (8, 256, 512, 65, 64, 1, 1, 0), (8, 256, 512, 8, 1, 5, 2, 0), (8, 256, 512, 64, 1, 5, 1, 0), (8, 32, 64, 128, 128, 3, 1, 0), (2, 20, 36, 24, 40, 3, 1, 0)]
for (B, H, W, Cin, Cout, k, dil, pro) in SHAPES:
    ldx, ldy = (Cin + 7) & ~7, (Cout + 7) & ~7
    x = torch.randn(B, H, W, ldx, device="cuda").to(dtype)[..., :Cin].permute(0, 3, 1, 2)
    g = (torch.randn(B, H, W, ldy, device="cuda") * 0.1).to(dtype)[..., :Cout].permute(0, 3, 1, 2)
    w = torch.zeros(Cout, Cin, k, k, device="cuda")
    bias = torch.zeros(Cout, device="cuda")
    sc = (torch.rand(1, Cin, device="cuda") + 0.5) if pro else None
    sh = (torch.rand(1, Cin, device="cuda") - 0.5) if pro else None
    pad = (k // 2) * dil
    spec = ops.ConvSpec('conv', k, k, 1, dil, pad, pad, H, W)
    res, tms = [], []
    for gen in ("1", ""):
        if gen: os.environ["SDHIP_WGRAD_GENERIC"] = gen
        else: os.environ.pop("SDHIP_WGRAD_GENERIC", None)
        __import__("pmt_learning_for_semantic_segmentation_and_disparity_amd")._lib.reload_diag()
        def go():
            return ops._wgrad_impl(x, ldx, g, ldy, w, bias, spec, sc, sh, bool(pro), 1)
        for _ in range(2): gw, gb = go()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): go()
        e1.record(); e1.synchronize()
        tms.append(e0.elapsed_time(e1) / 10 * 1e3)
        res.append((gw.float().clone(), gb.float().clone()))
    (a, ab), (b, bb) = res
    rel = ((a - b).norm() / (a.norm() + 1e-20)).item()
    relb = ((ab - bb).norm() / (ab.norm() + 1e-20)).item()
    ideal = max(2.0 * B * H * W * Cin * Cout * k * k / 2.5e15, 2.0 * B * H * W * (Cin + Cout) / 8e12) * 1e6
    print((B, H, W, Cin, Cout, k, dil, pro), "ideal_us %.1f generic %.1f fast %.1f  rel_dw %.2e rel_db %.2e" % (ideal, tms[0], tms[1], rel, relb), flush=True)
