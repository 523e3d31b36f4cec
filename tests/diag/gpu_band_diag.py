"""Where does the band kernel differ from the halo-tile kernel?  Prints the mismatch pattern per (pixel row mod 16, pixel
column mod 32, channel) for one shape; SDHIP_TUNE_BAND_DBG selects the kernel variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops, _lib

B, Ci, Co, H, W = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (4, 32, 32, 96, 256))]
torch.manual_seed(0)
x = torch.randn(B, H, W, Ci, device="cuda").bfloat16().permute(0, 3, 1, 2)
w = torch.randn(Co, Ci, 5, 5, device="cuda") * 0.05
b = torch.randn(Co, device="cuda")
def run():
    y = ops.conv2d(x, w, b, padding='same', act=0)
    torch.cuda.synchronize()
    return y.float().cpu()
y1 = run()
os.environ["SDHIP_CONV_NO_BAND"] = "1"; _lib.reload_diag()
y0 = run()
d = (y1 - y0).abs()
bad = d > 0.05 * y0.abs().max()
print("max diff %.4f of %.3f, bad fraction %.5f" % (d.max().item(), y0.abs().max().item(), bad.float().mean().item()))
if bad.any():
    idx = bad.nonzero()
    print("bad images", sorted(set(idx[:, 0].tolist())))
    print("bad channels", sorted(set(idx[:, 1].tolist())))
    print("bad rows mod 16", sorted(set((idx[:, 2] % 16).tolist())), " rows/16", sorted(set((idx[:, 2] // 16).tolist())))
    print("bad cols mod 32", sorted(set((idx[:, 3] % 32).tolist())), " cols/32", sorted(set((idx[:, 3] // 32).tolist())))
    print("first", idx[:5].tolist(), y1[tuple(idx[0])].item(), y0[tuple(idx[0])].item())
