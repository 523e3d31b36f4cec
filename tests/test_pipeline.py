"""The callers either side of the training step wired together: files on disk -> GPU-side sample
preparation -> training step with in-step metrics -> reference-format checkpoint -> resume."""
import os

import numpy as np
import pytest
import torch

from oracle import data_ref as DR
from oracle import ref_models as R
from oracle.detweights import fill_state_dict


def _write_sample(d, i, H, W, rng):
    from PIL import Image
    left = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    right = np.roll(left, 3, axis=1)
    seg = np.zeros((H, W, 3), dtype=np.uint8)
    seg[H // 4: 3 * H // 4, W // 3: 2 * W // 3] = 255                      # a "branch" blob
    depth = rng.uniform(2.0, 20.0, (H, W)).astype(np.float32)
    p = lambda n: os.path.join(d, "%s_%d" % (n, i))
    Image.fromarray(left).save(p("l") + ".png")
    Image.fromarray(right).save(p("r") + ".png")
    Image.fromarray(seg).save(p("s") + ".png")
    with open(p("d") + ".pfm", "wb") as f:
        f.write(b"Pf\n%d %d\n-1.0\n" % (W, H) + np.flipud(depth).tobytes())
    return p("l") + ".png", p("r") + ".png", p("s") + ".png", p("d") + ".pfm"


@pytest.mark.gpu
def test_files_to_checkpoint_round_trip(tmp_path):
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import checkpoint as C, nn as N, ops
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.data import SamplePreparer, draw_crop, read_sample_files
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.metrics import StepMetrics
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep
    rng = np.random.default_rng(0)
    H, W, B, crop = 288, 320, 2, (256, 256)
    files = [_write_sample(str(tmp_path), i, H, W, rng) for i in range(B)]
    norm = np.array([[0, 0, 0], [1, 1, 1]], dtype=np.float32)
    sp = SamplePreparer("roses", 2, 192, "linear", norm, dtype=torch.float32, device="cuda:0")
    batch = sp.alloc_batch(B, *crop)
    torch.manual_seed(7)
    crops = []
    for b, f in enumerate(files):
        left, right, seg, depth = read_sample_files(*f)
        c = draw_crop(H, W, crop, "roses")
        crops.append(c)
        sp.prepare_into(batch, b, left, right, seg, depth, c)
    torch.cuda.synchronize()
    sp.release()
    # slot 1 equals the oracle's preparation of the same files
    l1, r1, s1, d1 = read_sample_files(*files[1])
    want = DR.prepare_sample(l1, r1, s1, d1, "roses", 2, 192.0, "linear", norm, crops[1])
    for t, w in zip(batch, want):
        np.testing.assert_array_equal(t[1].cpu().numpy(), w)
    left_t, right_t, seg_t, disp_t = batch
    model = fill_state_dict(N.minidsnetExt(R.CFG(), labels=2, patch_type='1dcorr'), 5).cuda().train()
    metrics = StepMetrics(2, max_disp=1.0, device="cuda:0")
    ts = TrainStep(model, dtype=torch.float32, use_graph=False, metrics=metrics)
    losses = [float(ts(left_t, right_t, seg_t, disp_t)) for _ in range(3)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    out = metrics.compute()
    assert out["conf_matrix"].sum() == 3 * B * crop[0] * crop[1] and out["val_pxl"] == 3 * B * crop[0] * crop[1]
    assert 0 <= out["pixelAcc"] <= 1 and out["dispRMSE"] > 0
    path = C.save_checkpoint(C.make_state(ts, epoch=1, histories={"epoch_history": [1]}), 0.0, out["pixelAcc"], 9.0, 1.0, str(tmp_path / "run"))
    fresh = TrainStep(fill_state_dict(N.minidsnetExt(R.CFG(), labels=2, patch_type='1dcorr'), 6).cuda().train(), dtype=torch.float32,
                      use_graph=False)
    start, best, ehist = C.load_checkpoint_and_params(path, fresh, map_location="cuda:0")[:3]
    ops.set_step_context(None)
    assert start == 1 and ehist == [1] and best == [1.0, round(out["pixelAcc"], 4)]
    assert torch.equal(fresh.flat_p, ts.flat_p) and fresh.steps_done == 3


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus N` with no launcher environment starts N rank processes itself (the counterpart of
    mp.spawn(runNetwork, nprocs), torch_implementation.py:975) with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set;
    under a launcher (WORLD_SIZE present) it does not spawn again, and a --gpus / WORLD_SIZE mismatch is an error."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--rank-probe"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert sorted(x["rank"] for x in rows) == [0, 1, 2] and all(x["world"] == 3 for x in rows)
    assert all(x["rank"] == x["local_rank"] for x in rows)
    assert len({x["master"] for x in rows}) == 1 and rows[0]["master"].startswith("127.0.0.1:")
    assert len({x["pid"] for x in rows}) == 3 and len({x["ppid"] for x in rows}) == 1
    # launched by torch.distributed.run (the driver's N > 1 command): the rank runs in place
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rank-probe"],
                       env=dict(env, RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="1"),
                       capture_output=True, text=True, timeout=120)
    rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(rows) == 1 and rows[0]["rank"] == 1
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--rank-probe"],
                       env=dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 4" in r.stderr
