"""Launch one conv shape alone (for rocprofv3 runs): B H W Cin Cout k [prologue] [n]."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
B, H, W, Cin, Cout, k = [int(v) for v in sys.argv[1:7]]
pro = int(sys.argv[7]) if len(sys.argv) > 7 else 0
n = int(sys.argv[8]) if len(sys.argv) > 8 else 20
dtype = torch.bfloat16
x = torch.randn(B, H, W, Cin, device="cuda").to(dtype).permute(0, 3, 1, 2)
w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.03
wp = ops.packed_weight(w, 'conv', 'fwd', dtype)
y = ops.empty_nhwc(B, Cout, H, W, dtype, "cuda")
st = torch.zeros(ops.NREP, 2, 2, Cout, dtype=torch.float64, device="cuda")
sc = torch.rand(2, Cin, device="cuda") if pro else None
sh = torch.rand(2, Cin, device="cuda") if pro else None
def go():
    ops._conv_launch(x, Cin, wp, y, Cout, None, sc, sh, st, B, H, W, Cin, H, W, Cout, k, k, 1, 1, k // 2, k // 2, bool(pro), 2, 0, False, ops.NREP)
for _ in range(3): go()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): go()
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / n
print("shape", sys.argv[1:7], "pro", pro, "us/launch %.1f" % (ms * 1e3), "TFLOP/s %.1f" % (2.0 * B * H * W * Cin * Cout * k * k / ms / 1e9))
