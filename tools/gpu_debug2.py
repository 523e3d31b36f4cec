import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
dtype = torch.bfloat16
B, H, W, Cin, Cout, k = 1, 8, 32, 128, 64, 1
for c in (0, 8, 40, 64, 72, 100, 127):
    x = torch.zeros(B, H, W, Cin, device="cuda"); x[..., c] = 1.0
    x = x.to(dtype).permute(0, 3, 1, 2)
    w = torch.zeros(Cout, Cin, k, k, device="cuda")
    for ci in range(Cin): w[:, ci] = ci + 1
    w = w + torch.arange(Cout, device="cuda").view(-1, 1, 1, 1) * 0.0
    wp = ops.packed_weight(w, 'conv', 'fwd', dtype)
    y = ops.empty_nhwc(B, Cout, H, W, dtype, "cuda")
    os.environ["SDHIP_CONV_BIG"] = "1"
    ops._conv_launch(x, Cin, wp, y, Cout, None, None, None, None, B, H, W, Cin, H, W, Cout, k, k, 1, 1, 0, 0, False, 1, 0, False, 1)
    torch.cuda.synchronize()
    yy = y.float()
    print("c", c, "expect", c + 1, "got unique", torch.unique(yy).tolist()[:10], "count wrong", int((yy != c + 1).sum()))
