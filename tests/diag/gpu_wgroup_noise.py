"""Run-to-run noise of the bf16 step's flat gradient vs the difference grouped / per-layer weight gradients make."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import ref_models as R
from oracle.detweights import fill_state_dict
from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N, ops
from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch

def model():
    return fill_state_dict(N.minidsnetExt(R.CFG(aspp=0), labels=2, patch_type='1dcorr'), 31).cuda().train()

batch = synthetic_batch(2, 256, 256)
gs = []
for defer in (False, False, True, True):
    ts = TrainStep(model(), dtype=torch.bfloat16, use_graph=False, lr=0.0)
    ts.ctx.defer_wgrad = defer
    for _ in range(3):
        loss = ts(*batch)
    gs.append(ts.flat_g.clone()); print(defer, float(loss))
    ops.set_step_context(None)
n = float(gs[0].norm())
for i in range(4):
    for j in range(i + 1, 4):
        print(i, j, float((gs[i] - gs[j]).norm()) / n)
