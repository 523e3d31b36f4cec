"""Which host-side action between two replays of a captured TrainStep changes the next replay's loss?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import ref_models as R
from oracle.detweights import fill_state_dict
from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N, ops, checkpoint as C
from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch

batch = synthetic_batch(2, 256, 256)
mk = lambda seed: fill_state_dict(N.minidsnetExt(R.CFG(), labels=2, patch_type='1dcorr'), seed).cuda().train()
ref = TrainStep(mk(5), dtype=torch.float32, use_graph=False)
want = [float(ref(*batch)) for _ in range(12)]
print("eager:", ["%.4f" % v for v in want], flush=True)
ops.set_step_context(None)
del ref

def run(tag, action, lovasz=True):
    a = TrainStep(mk(5), dtype=torch.float32, use_graph=True, use_lovasz=lovasz)
    got = [float(a(*batch)) for _ in range(3)]          # steps 3, 4, 5
    keep = action(a)
    got.append(float(a(*batch)))                        # step 6
    got.append(float(a(*batch)))                        # step 7
    print("%-28s %s   (eager steps 3..7: %s)" % (tag, ["%.4f" % v for v in got], ["%.4f" % v for v in want[2:7]]), flush=True)
    ops.set_step_context(None)
    del a, keep

run("nothing", lambda a: None)
run("alloc 4 GB of NaN", lambda a: [torch.full((1 << 28,), float("nan"), device="cuda") for _ in range(4)])
run("empty_cache", lambda a: torch.cuda.empty_cache())
run("make_state", lambda a: C.make_state(a, epoch=1))
run("second model .cuda()", lambda a: mk(6))
run("second TrainStep", lambda a: TrainStep(mk(6), dtype=torch.float32, use_graph=False))
run("rng_reseed only", lambda a: ops.rng_reseed(a.flat_p.device, 0))
run("invalidate packs", lambda a: ops.invalidate_packed_weights())
run("second TrainStep, no lovasz", lambda a: TrainStep(mk(6), dtype=torch.float32, use_graph=False), lovasz=False)
