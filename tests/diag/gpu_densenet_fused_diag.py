"""bf16 DenseNet gradients: fused chain vs unfused chain, each against the f32 run of the same network."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle.detweights import fill_state_dict, rand_input
from pmt_learning_for_semantic_segmentation_and_disparity_amd import _lib as L
from pmt_learning_for_semantic_segmentation_and_disparity_amd.densenet import densenet121
x32 = rand_input(31, "img", (8, 3, 128, 256)).cuda()
wts = [0.7, 1.1, 0.9, 1.3, 0.8]
def run(dtype, fused):
    L.DIAG_NO_BNPRO = L.DIAG_NO_BNBWD_EPILOGUE = not fused
    m = fill_state_dict(densenet121(), 21).cuda().train()
    taps = m(x32.to(dtype), groups=2)
    sum(w * (t.float() * t.float()).mean() for w, t in zip(wts, taps)).backward()
    torch.cuda.synchronize()
    return {k: p.grad.float().clone() for k, p in m.named_parameters() if p.grad is not None}
g32 = run(torch.float32, False)
g1 = run(torch.bfloat16, True)
g0 = run(torch.bfloat16, False)
rel = lambda a, b: float(torch.linalg.norm(a - b)) / max(float(torch.linalg.norm(b)), 1e-12)
e1 = {k: rel(g1[k], g32[k]) for k in g32}
e0 = {k: rel(g0[k], g32[k]) for k in g32}
d10 = {k: rel(g1[k], g0[k]) for k in g32}
srt = lambda d: sorted(d.values())
n = len(g32)
print("tensors", n)
print("vs f32:  fused   median %.4f  p90 %.4f  max %.4f" % (srt(e1)[n // 2], srt(e1)[int(n * .9)], srt(e1)[-1]))
print("vs f32:  unfused median %.4f  p90 %.4f  max %.4f" % (srt(e0)[n // 2], srt(e0)[int(n * .9)], srt(e0)[-1]))
print("fused vs unfused: median %.4f p90 %.4f max %.4f" % (srt(d10)[n // 2], srt(d10)[int(n * .9)], srt(d10)[-1]))
for k in sorted(d10, key=lambda k: -d10[k])[:8]:
    print("  %-50s fused/unf %.3f  fused/f32 %.3f  unf/f32 %.3f  |g| %.3e" % (k, d10[k], e1[k], e0[k], float(torch.linalg.norm(g32[k]))))
