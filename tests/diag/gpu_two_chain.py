"""Does splitting the two DenseNet towers into two concurrent stream chains beat the batched single chain?  (timing only)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N, ops
from pmt_learning_for_semantic_segmentation_and_disparity_amd.densenet import densenet121

torch.manual_seed(0)
m = densenet121().cuda().train()
B, H, W = 8, 256, 512
x = torch.zeros(2 * B, H, W, 8, device="cuda", dtype=torch.bfloat16)
x[..., :3] = torch.rand(2 * B, H, W, 3, device="cuda")
both = x.permute(0, 3, 1, 2)
left, right = both[:B], both[B:]
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

def batched(grad):
    t = m(both, groups=2)
    if grad:
        sum(u.float().mean() for u in t).backward()

def split(grad):
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur); sb.wait_stream(cur)
    with torch.cuda.stream(sa):
        ta = m(left, groups=1)
        la = sum(u.float().mean() for u in ta) if grad else None
    with torch.cuda.stream(sb):
        tb = m(right, groups=1)
        lb = sum(u.float().mean() for u in tb) if grad else None
    if grad:
        with torch.cuda.stream(sa):
            la.backward()
        with torch.cuda.stream(sb):
            lb.backward()
    cur.wait_stream(sa); cur.wait_stream(sb)

for grad in (False, True):
    for name, fn in (("batched one chain", batched), ("two chains", split)):
        ops.set_step_context(None)
        for _ in range(2):
            m.zero_grad(set_to_none=True); fn(grad)
        torch.cuda.synchronize()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            m.zero_grad(set_to_none=True); fn(grad)
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        m.zero_grad(set_to_none=True)
        with torch.cuda.graph(g):
            fn(grad)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): g.replay()
        e1.record(); e1.synchronize()
        print("DenseNet towers %s, %s: %.3f ms" % ("fwd+bwd" if grad else "fwd", name, e0.elapsed_time(e1) / 10), flush=True)
        del g
