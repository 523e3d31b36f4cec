"""Cost of one dependent kernel node in a replayed hipGraph (tiny kernels, one stream): lower bound per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
x = torch.randn(2, 64, 16, 16, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
for n in (200, 1000):
    def body():
        y = x
        for _ in range(n):
            y = ops.affine_act(y, None, None, None, 1)
        return y
    with torch.no_grad():
        body(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): g.replay()
        e1.record(); e1.synchronize()
        print("graph of %d dependent tiny kernels: %.2f us per node" % (n, e0.elapsed_time(e1) / 5 / n * 1e3))
        e0.record()
        for _ in range(5): body()
        e1.record(); e1.synchronize()
        print("eager %d: %.2f us per launch" % (n, e0.elapsed_time(e1) / 5 / n * 1e3))
