"""Multi-rank code paths on ONE process: world_size 2 with an all-reduce that doubles (both 'ranks' hold the same shard)
vs one rank on the duplicated batch.  Per-module relative gradient differences."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ref_models as R
from oracle.detweights import fill_state_dict
from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N, parallel
from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch
shard = synthetic_batch(2, 256, 256, seed=77)
dup = [torch.cat([t, t], 0) for t in shard]
def run(world):
    m = fill_state_dict(N.minidsnetExt(R.CFG(aspp=0), labels=2, patch_type='1dcorr'), 51).cuda().train()
    step = TrainStep(m, dtype=torch.float32, use_graph=False, use_lovasz=False, world_size=world)
    if world > 1:
        parallel._state["world"] = world
        parallel.all_reduce_sum_ = lambda t: t.mul_(world)
    loss = step.forward_backward(*(shard if world > 1 else dup))
    if world > 1: step.flat_g.mul_(world)      # the gradient all-reduce
    torch.cuda.synchronize()
    return m, float(loss), (step.flat_g / world).cpu().numpy().copy()
m, l1, g1 = run(1)
_, l2, g2 = run(2)
print("loss 1-rank %.6f fake-2-rank %.6f total rel %.3e" % (l1, l2, np.linalg.norm(g2 - g1) / np.linalg.norm(g1)))
off = 0; grp = collections.defaultdict(list)
for name, p in m.named_parameters():
    n = p.numel(); a, b = g1[off:off + n], g2[off:off + n]; off += n
    grp[name.split(".")[0]].append(np.linalg.norm(a - b) / max(np.linalg.norm(a), 1e-20))
for k, v in grp.items(): print("%-24s n=%3d median rel %.3e max %.3e" % (k, len(v), np.median(v), max(v)))
