"""Time conv_fwd shapes: general kernel (SDHIP_CONV_GENERIC=1) vs fast path; graph-replayed to exclude launch overhead."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
dtype = torch.bfloat16
SHAPES = [(8, 256, 512, 64, 65, 1, 1), (8, 256, 512, 65, 64, 1, 1), (8, 256, 512, 32, 65, 1, 1), (8, 256, 512, 1, 64, 5, 1), (8, 256, 512, 64, 64, 5, 1), (8, 256, 512, 32, 32, 3, 1), (8, 256, 512, 64, 64, 3, 1), (8, 256, 512, 64, 64, 1, 1),
          (8, 256, 512, 8, 1, 5, 2), (8, 256, 512, 72, 64, 1, 1), (8, 256, 512, 64, 72, 1, 1),
          (16, 64, 128, 128, 32, 3, 1), (16, 64, 128, 32, 128, 3, 1), (8, 64, 128, 64, 64, 3, 1), (16, 32, 64, 128, 32, 3, 1),
          (16, 16, 32, 128, 32, 3, 1), (16, 32, 64, 256, 128, 1, 1), (16, 16, 32, 1024, 128, 1, 1), (16, 64, 128, 128, 224, 1, 1)]
for (B, H, W, Cin, Cout, k, dil) in SHAPES:
    ldx = (Cin + 7) & ~7
    x = torch.randn(B, H, W, ldx, device="cuda").to(dtype)[..., :Cin].permute(0, 3, 1, 2)
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.03
    wp = ops.packed_weight(w, 'conv', 'fwd', dtype)
    y, ldy = ops.alloc_nhwc(B, Cout, H, W, dtype, "cuda")
    st = torch.zeros(ops.NREP, 2, 2, Cout, dtype=torch.float64, device="cuda")
    pad = (k // 2) * dil
    def go():
        ops._conv_launch(x, ldx, wp, y, ldy, None, None, None, st, B, H, W, Cin, H, W, Cout, k, k, 1, dil, pad, pad, False, 2, 0, False, ops.NREP)
    out = []
    for gen in ("1", ""):
        if gen: os.environ["SDHIP_CONV_GENERIC"] = gen
        else: os.environ.pop("SDHIP_CONV_GENERIC", None)
        __import__("pmt_learning_for_semantic_segmentation_and_disparity_amd")._lib.reload_diag()
        for _ in range(3): go()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): go()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        out.append(e0.elapsed_time(e1) / 20 * 1e3)
        if gen: yref = y.clone()
    err = (y.float() - yref.float()).abs().max().item()
    ideal = max(2.0 * B * H * W * Cin * Cout * k * k / 2.5e15, 2.0 * B * H * W * (Cin + Cout) / 8e12) * 1e6
    print((B, H, W, Cin, Cout, k, dil), "ideal_us %.1f generic %.1f fast %.1f  maxdiff %.3g" % (ideal, out[0], out[1], err), flush=True)
