"""5x5 convolution at the bench shape: persistent band kernel (conv_band.h) vs the halo-tile kernel
(SDHIP_CONV_NO_BAND=1), HIP-event timed over back-to-back launches; prints max |difference| of the outputs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops, _lib

def run(B, H, W, Ci, Co, stats, n=20, k=5):
    dtype = torch.bfloat16
    x = torch.randn(B, H, W, Ci, device="cuda").to(dtype).permute(0, 3, 1, 2)
    w = torch.randn(Co, Ci, k, k, device="cuda") * 0.03
    wp = ops.packed_weight(w, 'conv', 'fwd', dtype)
    y = ops.empty_nhwc(B, Co, H, W, dtype, "cuda")
    st = torch.zeros(ops.NREP, 1, 2, Co, dtype=torch.float64, device="cuda") if stats else None
    def go():
        ops._conv_launch(x, Ci, wp, y, Co, None, None, None, st, B, H, W, Ci, H, W, Co, k, k, 1, 1, k // 2, k // 2, False, 1, 0, False, ops.NREP)
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): go()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, y.float().clone(), (st.clone() if stats else None)

if len(sys.argv) > 1 and sys.argv[1] == "--k3":   # 3x3, <= 32 input channels: band kernel vs halo-tile kernel (SDHIP_CONV_NO_BAND3)
    for (B, H, W, Ci, Co) in [(8, 256, 512, 32, 32), (8, 256, 512, 32, 64), (8, 128, 256, 32, 32), (16, 256, 512, 32, 32)]:
        os.environ.pop("SDHIP_CONV_NO_BAND3", None); _lib.reload_diag()
        torch.manual_seed(1); t1, y1, _ = run(B, H, W, Ci, Co, False, k=3)
        os.environ["SDHIP_CONV_NO_BAND3"] = "1"; _lib.reload_diag()
        torch.manual_seed(1); t0, y0, _ = run(B, H, W, Ci, Co, False, k=3)
        print("3x3 B%d %dx%d %d->%d: band %.1f us  old %.1f us  max|dy| %.4f  mismatch %.5f" % (B, H, W, Ci, Co, t1, t0, (y1 - y0).abs().max().item(), (y1 != y0).float().mean().item()), flush=True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "--dbg":   # timing only, band path, SDHIP_TUNE_BAND_DBG taken from the environment
    for (B, H, W, Ci, Co) in [(8, 256, 512, 64, 64), (8, 256, 512, 32, 32)]:
        t1, _, _ = run(B, H, W, Ci, Co, False, n=30)
        print("dbg=%s B%d %d->%d: %.1f us" % (os.environ.get("SDHIP_TUNE_BAND_DBG", "0"), B, Ci, Co, t1), flush=True)
    sys.exit(0)
for (B, H, W, Ci, Co) in [(8, 256, 512, 64, 64), (8, 256, 512, 32, 32), (16, 256, 512, 32, 32), (8, 256, 512, 64, 32)]:
    for stats in (False, True):
        os.environ.pop("SDHIP_CONV_NO_BAND", None); _lib.reload_diag()
        torch.manual_seed(1)
        t1, y1, s1 = run(B, H, W, Ci, Co, stats)
        os.environ["SDHIP_CONV_NO_BAND"] = "1"; _lib.reload_diag()
        torch.manual_seed(1)
        t0, y0, s0 = run(B, H, W, Ci, Co, stats)
        flops = 2.0 * B * H * W * Ci * Co * 25
        print("B%d %dx%d %d->%d stats=%d: band %.1f us (%.0f TF/s)  old %.1f us (%.0f TF/s)  max|dy| %.4f of %.2f  mismatch %.4f" % (
            B, H, W, Ci, Co, stats, t1, flops / t1 * 1e-6, t0, flops / t0 * 1e-6, (y1 - y0).abs().max().item(), y0.abs().max().item(),
            (y1 != y0).float().mean().item()), flush=True)
        if stats:
            a, b = s1.sum(0), s0.sum(0)
            print("   stats rel diff %.2e" % ((a - b).abs().max() / b.abs().max()).item(), flush=True)
