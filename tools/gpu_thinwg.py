"""Thin (8 -> 1 channel, 5x5 dilation 2) weight gradient at 8 x 256 x 512 alone, HIP-event timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
B, H, W = 8, 256, 512
dev = torch.device("cuda:0")
x = torch.randn(B, H, W, 8, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2)
g1 = (torch.randn(B, H, W, 1, device=dev) * 0.1).to(torch.bfloat16).permute(0, 3, 1, 2)
w = torch.zeros(1, 8, 5, 5, device=dev)
spec = ops.ConvSpec('conv', 5, 5, 1, 2, 4, 4, H, W)
from pmt_learning_for_semantic_segmentation_and_disparity_amd import _lib
acc = torch.zeros(_lib.packed_elems(1, 8, 25, _lib.BF16), dtype=torch.float32, device=dev)
def launch():
    _lib.call("sdhip_conv2d_wgrad", _lib.ptr(x), _lib.ptr(g1), _lib.ptr(acc), None, None, None, B, H, W, 8, 8, H, W, 1, 1, 5, 5, 1, 2, 4, 4,
              1, 1, 1, 1, 0, 0, 1, 1, _lib.BF16, _lib.stream_ptr())
for _ in range(3):
    launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    launch()
e1.record()
torch.cuda.synchronize()
print("blocks cap %s: %.1f us" % (os.environ.get("SDHIP_TUNE_THIN_BLOCKS", "256"), e0.elapsed_time(e1) * 20))
