"""Per-parameter gradient difference: 2 ranks x 2 pairs (gloo, shared GPU) vs 1 rank x 4 pairs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import torch.multiprocessing as mp
from test_parallel import _gpu_rank_worker
if __name__ == "__main__":
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = 29611
    procs = [ctx.Process(target=_gpu_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=500) for _ in procs], key=lambda r: r[0])
    [p.join(60) for p in procs]
    from oracle import ref_models as R
    from oracle.detweights import fill_state_dict
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch
    m = fill_state_dict(N.minidsnetExt(R.CFG(aspp=0), labels=2, patch_type='1dcorr'), 51).cuda().train()
    step = TrainStep(m, dtype=torch.float32, use_graph=False, use_lovasz=False)
    loss = step.forward_backward(*synthetic_batch(4, 256, 256, seed=77))
    torch.cuda.synchronize()
    g1 = step.flat_g.cpu().numpy(); g2 = res[0][2]
    print("loss", float(loss), res[0][1], res[1][1], "total rel", np.linalg.norm(g2 - g1) / np.linalg.norm(g1))
    off = 0; rows = []
    for name, p in m.named_parameters():
        n = p.numel(); a, b = g1[off:off + n], g2[off:off + n]; off += n
        rows.append((np.linalg.norm(a - b), np.linalg.norm(a), name))
    rows.sort(key=lambda r: r[0] / max(r[1], 1e-20))
    for d, nrm, name in rows[:12]: print("%-70s |diff| %.3e |g| %.3e rel %.3e" % (name, d, nrm, d / max(nrm, 1e-20)))
    print("...")
    for d, nrm, name in rows[-12:]: print("%-70s |diff| %.3e |g| %.3e rel %.3e" % (name, d, nrm, d / max(nrm, 1e-20)))
    import collections
    grp = collections.defaultdict(list)
    for d, nrm, name in rows: grp[name.split(".")[0]].append(d / max(nrm, 1e-20))
    for k, v in grp.items(): print("%-24s n=%3d median rel %.3e max %.3e" % (k, len(v), np.median(v), max(v)))
