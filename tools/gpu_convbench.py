"""Launch the dominant conv kernel alone (for rocprofv3 --pmc / --kernel-trace runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
B, H, W, C = 8, 256, 512, 64
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dtype = torch.bfloat16
x = torch.randn(B, H, W, C, device="cuda").to(dtype).permute(0, 3, 1, 2)
w = torch.randn(C, C, k, k, device="cuda") * 0.03
wp = ops.packed_weight(w, 'conv', 'fwd', dtype)
y = ops.empty_nhwc(B, C, H, W, dtype, "cuda")
st = torch.zeros(ops.NREP, 1, 2, C, dtype=torch.float64, device="cuda")
for _ in range(n):
    ops._conv_launch(x, C, wp, y, C, None, None, None, st, B, H, W, C, H, W, C, k, k, 1, 1, k // 2, k // 2, False, 1, 0, False, ops.NREP)
torch.cuda.synchronize()
print("done")
