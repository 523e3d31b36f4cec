"""1x1 convolutions: streaming GEMM (conv_gemm.h) vs the halo-tile kernel (SDHIP_CONV_NO_GEMM=1), graph-replayed;
forward with BatchNorm prologue + statistics (DenseNet conv1), plain data-gradient form, decoder 1x1s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops, _lib
dtype = torch.bfloat16
# (B, H, W, Cin, ldx, Cout, prologue+stats)
SHAPES = [(16, 64, 128, 64, 256, 128, 1), (16, 64, 128, 224, 256, 128, 1), (16, 64, 128, 128, 128, 224, 0), (16, 64, 128, 256, 256, 128, 1),
          (16, 32, 64, 128, 512, 128, 1), (16, 32, 64, 480, 512, 128, 1), (16, 32, 64, 128, 128, 480, 0), (16, 32, 64, 512, 512, 256, 1),
          (16, 16, 32, 256, 1024, 128, 1), (16, 16, 32, 992, 1024, 128, 1), (16, 16, 32, 128, 128, 992, 0), (16, 16, 32, 1024, 1024, 512, 1),
          (16, 8, 16, 512, 1024, 128, 1), (16, 8, 16, 992, 1024, 128, 1), (16, 8, 16, 128, 128, 992, 0),
          (8, 256, 512, 65, 72, 64, 0), (8, 256, 512, 64, 64, 65, 0), (8, 256, 512, 33, 40, 32, 0), (8, 64, 128, 512, 512, 128, 0), (8, 16, 32, 2048, 2048, 64, 0)]
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
    out = []
    for old in ("1", ""):
        if old: os.environ["SDHIP_CONV_NO_GEMM"] = old
        else: os.environ.pop("SDHIP_CONV_NO_GEMM", None)
        _lib.reload_diag()
        for _ in range(3): go()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): go()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        out.append(e0.elapsed_time(e1) / 20 * 1e3)
        if old: yref = y.clone()
    err = (y.float() - yref.float()).abs().max().item()
    nbytes = 2.0 * B * H * W * (Cin + Cout)
    print((B, H, W, Cin, Cout, "pro" if pro else "plain"), "hbm-ideal_us %5.1f  halo-tile %6.1f  gemm %6.1f  (%.2f TB/s)  maxdiff %.3g" % (
        nbytes / 8e12 * 1e6, out[0], out[1], nbytes / out[1] / 1e6, err), flush=True)
