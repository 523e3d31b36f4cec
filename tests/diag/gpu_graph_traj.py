"""Eager vs hipGraph loss trajectories at the shipped learning rate (diagnostic for the checkpoint-resume test)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import ref_models as R
from oracle.detweights import fill_state_dict
from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N, ops
from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch

batch = synthetic_batch(2, 256, 256)
mk = lambda: fill_state_dict(N.minidsnetExt(R.CFG(), labels=2, patch_type='1dcorr'), 5).cuda().train()
for graph in (False, True, False, True):
    ts = TrainStep(mk(), dtype=torch.float32, use_graph=graph)
    ls = [float(ts(*batch)) for _ in range(6 if not graph else 4)]
    print("graph" if graph else "eager", ["%.4f" % l for l in ls], "steps_done", ts.steps_done, flush=True)
    # loss of a fresh eager forward with the current parameters
    ops.set_step_context(None)
    with torch.no_grad():
        o = ts.model(batch[0], batch[1])
    print("   fresh forward loss at current params: %.4f" % float(ops.train_loss(o[0], o[1], o[2], batch[2], batch[3], True)), flush=True)
