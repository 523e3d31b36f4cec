"""Does a captured loss (CE + Lovasz + L1) survive unrelated small allocations between replays?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import synthetic_batch

left, right, seg, disp = synthetic_batch(2, 256, 256)
torch.manual_seed(0)
s1 = torch.randn(2, 2, 256, 256, device="cuda").contiguous(memory_format=torch.channels_last)
s2 = torch.randn(2, 2, 256, 256, device="cuda").contiguous(memory_format=torch.channels_last)
d = torch.rand(2, 1, 256, 256, device="cuda") * 8

def one(lov):
    a, b, c = (t.clone().requires_grad_(True) for t in (s1, d, s2))
    loss = ops.train_loss(a, b, c, seg, disp, lov)
    loss.backward()
    return loss.detach(), c.grad

for lov in (True, False):
    print("eager lovasz=%s: %.6f" % (lov, float(one(lov)[0])), flush=True)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        one(lov)
    torch.cuda.current_stream().wait_stream(st)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out, gc = one(lov)
    g.replay(); print("  replay 1: %.6f  |grad| %.6f" % (float(out), float(gc.float().norm())), flush=True)
    junk = [torch.randn(1000, device="cuda") for _ in range(3000)]
    g.replay(); print("  replay after 3000 small device allocations: %.6f  |grad| %.6f" % (float(out), float(gc.float().norm())), flush=True)
    junk2 = [torch.randn(1000).cuda() for _ in range(1500)]
    g.replay(); print("  replay after 1500 small H2D copies: %.6f  |grad| %.6f" % (float(out), float(gc.float().norm())), flush=True)
    big = torch.randn(1 << 24).cuda()
    g.replay(); print("  replay after one big H2D copy: %.6f  |grad| %.6f" % (float(out), float(gc.float().norm())), flush=True)
    del junk, junk2, big

sys.exit(0)
# ---- node inventory of the loss-only graph (what does rocPRIM put into a capture?)
a, b, c = (t.clone().requires_grad_(True) for t in (s1, d, s2))
st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    ops.train_loss(a, b, c, seg, disp, True)
torch.cuda.current_stream().wait_stream(st)
g = torch.cuda.CUDAGraph()
g.enable_debug_mode()
with torch.cuda.graph(g):
    out = ops.train_loss(a, b, c, seg, disp, True)
os.makedirs("gpurun_out", exist_ok=True)
g.debug_dump("gpurun_out/r2_lossgraph.dot")
txt = open("gpurun_out/r2_lossgraph.dot").read()
import re, collections
kinds = collections.Counter(re.findall(r'label="([A-Za-z_ ]+)', txt))
print("graph node labels:", dict(kinds), flush=True)
for line in txt.splitlines():
    if "MEMCPY" in line.upper() or "memcpy" in line or "MEMSET" in line.upper():
        print("   ", line.strip()[:300], flush=True)
