"""fused vs unfused BatchNorm paths on one rank: per-parameter gradient difference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ref_models as R
from oracle.detweights import fill_state_dict
from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N
from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch
gs = []
for flag in ("", "1"):
    if flag: os.environ["SDHIP_DIAG_NO_FUSED_BN"] = flag
    m = fill_state_dict(N.minidsnetExt(R.CFG(aspp=0), labels=2, patch_type='1dcorr'), 51).cuda().train()
    step = TrainStep(m, dtype=torch.float32, use_graph=False, use_lovasz=False)
    loss = step.forward_backward(*synthetic_batch(4, 256, 256, seed=77))
    torch.cuda.synchronize()
    gs.append((float(loss), step.flat_g.cpu().numpy().copy()))
g1, g2 = gs[0][1], gs[1][1]
print("loss fused %.6f unfused %.6f  total rel %.3e" % (gs[0][0], gs[1][0], np.linalg.norm(g2 - g1) / np.linalg.norm(g1)))
off = 0; rows = []
for name, p in m.named_parameters():
    n = p.numel(); a, b = g1[off:off + n], g2[off:off + n]; off += n
    rows.append((np.linalg.norm(a - b) / max(np.linalg.norm(a), 1e-20), np.linalg.norm(a), name))
rows.sort(key=lambda r: -r[0])
for r in rows[:12]: print("%-70s rel %.3e |g| %.3e" % (r[2], r[0], r[1]))
print("median rel", np.median([r[0] for r in rows]))
