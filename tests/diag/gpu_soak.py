"""Soak: N graph replays of the benchmark's training steps on a fixed synthetic batch — the loss must stay finite and fall
(python tests/diag/gpu_soak.py [steps]).  Uses bench.py's model / loss construction; no oracle involved."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch
from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dt = torch.bfloat16
for name, B, H, W in (("minidsnetExt", 8, 256, 512), ("dsnet", 8, 256, 512), ("psmnet", 8, 256, 512), ("minidsnetExt_cfg5", 4, 512, 1024)):
    model = bench.build_model(dt, name)
    loss_fn = None
    if name == "psmnet":
        loss_fn = lambda outs, seg, disp: ops.mean_l1_loss(outs, disp[:, 0])
    elif name == "dsnet":
        loss_fn = lambda outs, seg, disp: ops.train_loss(outs[0], outs[1], outs[2], seg, disp, True)
    elif name == "minidsnetExt_cfg5":
        loss_fn = lambda outs, seg, disp: ops.train_loss(outs[0], outs[1], outs[2], seg, disp, True, True, True)
    step = TrainStep(model, dtype=dt, use_graph=True, loss_fn=loss_fn, lr=1e-4)
    batch = synthetic_batch(B, H, W, labels=19 if name == "minidsnetExt_cfg5" else 2, seed=11)
    losses = []
    for i in range(N):
        l = step(*batch)
        if i % max(1, N // 6) == 0 or i == N - 1:
            losses.append(float(l.item()))
    ok = all(math.isfinite(v) for v in losses) and losses[-1] < losses[0]
    fin = all(bool(torch.isfinite(p).all()) for p in model.parameters())
    print("%-18s graph=%s  losses %s  finite params %s  %s" % (name, step.use_graph, ["%.4f" % v for v in losses], fin, "OK" if ok and fin else "FAIL"), flush=True)
    ops.set_step_context(None)
    del step, model, batch
    torch.cuda.empty_cache()
