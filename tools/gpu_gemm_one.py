"""One 1x1 shape on the streaming GEMM, 30 launches (for rocprofv3 counter passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops, _lib
dtype = torch.bfloat16
B, H, W, Cin, ldx, Cout, pro = [int(v) for v in sys.argv[1:8]]
x = torch.randn(B, H, W, ldx, device="cuda").to(dtype)[..., :Cin].permute(0, 3, 1, 2)
w = torch.randn(Cout, Cin, 1, 1, device="cuda") * 0.03
wp = ops.packed_weight(w, 'conv', 'fwd', dtype)
y, ldy = ops.alloc_nhwc(B, Cout, H, W, dtype, "cuda")
st = torch.zeros(ops.NREP, 2, 2, Cout, dtype=torch.float64, device="cuda") if pro else None
sc = (torch.rand(2, Cin, device="cuda") + 0.5) if pro else None
sh = (torch.rand(2, Cin, device="cuda") - 0.5) if pro else None
for _ in range(30):
    ops._conv_launch(x, ldx, wp, y, ldy, None, sc, sh, st, B, H, W, Cin, H, W, Cout, 1, 1, 1, 1, 0, 0, bool(pro), 2 if pro else 1, 0, False, ops.NREP)
torch.cuda.synchronize()
