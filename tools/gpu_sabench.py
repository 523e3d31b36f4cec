"""Soft-argmin head at the bench shape (B=8, 48 levels, 64x128 -> 192 x 256 x 512): kernels under rocprofv3 (tools/kprof.sh softargmin)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
B, D4, H4, W4 = 8, 48, 64, 128
c = torch.randn(B * D4, 1, H4, W4, device="cuda").bfloat16().requires_grad_(True)
for _ in range(5):
    p = ops.soft_argmin(c, D4, 192, 256, 512)
    p.backward(torch.randn_like(p))
torch.cuda.synchronize()
