"""Where does the streaming 1x1 GEMM lose its time?  SDHIP_TUNE_GEMM_DBG bits: 1 no stores, 2 weights once, 4 no MFMA."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops, _lib
dtype = torch.bfloat16
SHAPES = [(16, 64, 128, 224, 256, 128, 1), (16, 64, 128, 128, 128, 224, 0), (8, 256, 512, 64, 64, 65, 0), (8, 256, 512, 65, 72, 64, 0), (16, 16, 32, 992, 1024, 128, 1)]
for (B, H, W, Cin, ldx, Cout, pro) in SHAPES:
    x = torch.randn(B, H, W, ldx, device="cuda").to(dtype)[..., :Cin].permute(0, 3, 1, 2)
    w = torch.randn(Cout, Cin, 1, 1, device="cuda") * 0.03
    wp = ops.packed_weight(w, 'conv', 'fwd', dtype)
    y, ldy = ops.alloc_nhwc(B, Cout, H, W, dtype, "cuda")
    st = torch.zeros(ops.NREP, 2, 2, Cout, dtype=torch.float64, device="cuda") if pro else None
    sc = (torch.rand(2, Cin, device="cuda") + 0.5) if pro else None
    sh = (torch.rand(2, Cin, device="cuda") - 0.5) if pro else None
    def go():
        ops._conv_launch(x, ldx, wp, y, ldy, None, sc, sh, st, B, H, W, Cin, H, W, Cout, 1, 1, 1, 1, 0, 0, bool(pro), 2 if pro else 1, 0, False, ops.NREP)
    res = []
    for dbg in (0, 1, 5, 29):
        os.environ["SDHIP_TUNE_GEMM_DBG"] = str(dbg)
        _lib.reload_diag()
        for _ in range(3): go()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): go()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        res.append("dbg%d %.1f" % (dbg, e0.elapsed_time(e1) / 20 * 1e3))
    print((B, H, W, Cin, Cout, "pro" if pro else "plain"), "  ".join(res), flush=True)
