"""One weight-gradient shape alone (for rocprofv3): B H W Cin Cout k [prologue] [n]."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
B, H, W, Cin, Cout, k = [int(v) for v in sys.argv[1:7]]
pro = int(sys.argv[7]) if len(sys.argv) > 7 else 0
n = int(sys.argv[8]) if len(sys.argv) > 8 else 5
dtype = torch.bfloat16
x = torch.randn(B, H, W, Cin, device="cuda").to(dtype).permute(0, 3, 1, 2)
g = (torch.randn(B, H, W, Cout, device="cuda") * 0.1).to(dtype).permute(0, 3, 1, 2)
w = torch.zeros(Cout, Cin, k, k, device="cuda")
sc = (torch.rand(1, Cin, device="cuda") + 0.5) if pro else None
sh = (torch.rand(1, Cin, device="cuda") - 0.5) if pro else None
spec = ops.ConvSpec('conv', k, k, 1, 1, k // 2, k // 2, H, W)
for _ in range(n):
    ops._wgrad_impl(x, Cin, g, Cout, w, None, spec, sc, sh, bool(pro), 1)
torch.cuda.synchronize()
