"""Locate the error of y = conv + addend (band kernel) against an f32 reference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
from pmt_learning_for_semantic_segmentation_and_disparity_amd._lib import call, ptr, stream_ptr, dtype_code
B, ci, co, H, W = 2, 64, 64, 100, 450
g = torch.Generator().manual_seed(7 * B + ci)
x = torch.randn(B, H, W, ci, generator=g).cuda().bfloat16().permute(0, 3, 1, 2)
a = torch.randn(B, H, W, co, generator=g).cuda().bfloat16().permute(0, 3, 1, 2)
w = (torch.randn(co, ci, 5, 5, generator=g) * 0.05).cuda()
wp = ops.packed_weight(w, 'conv', 'fwd', torch.bfloat16)
conv = F.conv2d(x.float().cpu(), w.bfloat16().float().cpu(), None, padding=2)
ref = conv + a.float().cpu()
y = ops.empty_nhwc(B, co, H, W, torch.bfloat16, "cuda")
call("sdhip_conv2d_fwd_add", ptr(x), ptr(wp), ptr(y), ptr(a), co, B, H, W, ci, ci, H, W, co, co, 5, 5, 2, 2, dtype_code(x), stream_ptr())
y0 = ops.empty_nhwc(B, co, H, W, torch.bfloat16, "cuda")
ops._conv_launch(x, ci, wp, y0, co, None, None, None, None, B, H, W, ci, H, W, co, 5, 5, 1, 1, 2, 2, False, 1, 0, False)
torch.cuda.synchronize()
yc = y.float().cpu()
d = (yc - ref).abs()
print("max err", d.max().item(), "of", ref.abs().max().item(), "plain conv err", (y0.float().cpu() - conv).abs().max().item())
bad = d > 0.04
print("bad frac", bad.float().mean().item())
idx = bad.nonzero()
if len(idx):
    print("rows%16", sorted(set((idx[:, 2] % 16).tolist())), "cols%32", sorted(set((idx[:, 3] % 32).tolist())), "ch", sorted(set(idx[:, 1].tolist()))[:70])
    print("rows", sorted(set(idx[:, 2].tolist()))[:40], "cols>=", idx[:, 3].min().item(), idx[:, 3].max().item())
    i = tuple(idx[0].tolist()); print(i, yc[i].item(), ref[i].item(), conv[i].item(), a.float().cpu()[i].item())
