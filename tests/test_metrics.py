"""On-device step metrics.
CPU: the numpy oracle (oracle/metrics_ref.py) against tests/golden/metrics.npz, which holds the results of the
reference's own SegAccuracyNp / GetSegMetricsNp / unnormalizedErrorNP / GetDispMetricsNp (oracle/make_golden.py
gen_metrics).  GPU: `sdhip_step_metrics` (through metrics.StepMetrics) against the golden values and the oracle:
integer counters bit-exact, float scores to 1e-5 relative (f64 accumulation vs numpy's float32 pairwise mean)."""
import os

import numpy as np
import pytest
import torch

from oracle import metrics_ref as MR

GOLD = os.path.join(os.path.dirname(__file__), "golden", "metrics.npz")
CASES = ("roses", "city", "one")
SCALARS = ("pixelAcc", "pixelPrec", "pixelRecall", "pixelF1", "pixelBF1", "err", "val_pxl", "dispRMSE", "dispSqRel",
           "BdispRMSE", "BdispSqRel")
FTOL = 1e-5


def _case(gold, name):
    g = lambda k: gold["%s.%s" % (name, k)]
    return g("logits"), g("seg_full"), g("disp_pred"), g("disp"), int(g("labels")), float(g("max_disp")), bool(g("mask_invalid"))


def _same(a, b):
    a, b = float(a), float(b)
    if np.isnan(b) or np.isinf(b):
        return (np.isnan(a) and np.isnan(b)) or a == b
    return abs(a - b) <= FTOL * max(1.0, abs(b))


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_metrics(name):
    gold = np.load(GOLD)
    logits, seg_full, dp, dg, L, max_disp, mi = _case(gold, name)
    out = MR.step_metrics(logits, seg_full, dp, dg, L, max_disp, mi)
    np.testing.assert_array_equal(out["conf_matrix"], gold[name + ".conf_matrix"])
    for k in SCALARS:
        assert _same(out[k], gold["%s.%s" % (name, k)]), (k, out[k], gold["%s.%s" % (name, k)])


def _gpu_metrics(logits, seg_full, dp, dg, L, max_disp, mi, dtype=torch.float32, repeat=1):
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.metrics import StepMetrics
    dev = torch.device("cuda:0")
    m = StepMetrics(L, max_disp=max_disp, mask_invalid=mi, device=dev)
    t = lambda a, d: None if a is None else torch.from_numpy(a).to(dev).to(d)
    for _ in range(repeat):
        m.update(t(logits, dtype), t(seg_full, torch.float32), t(dp, dtype), t(dg, torch.float32))
    torch.cuda.synchronize()
    return m, m.compute()


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_metrics_match_reference(name):
    gold = np.load(GOLD)
    args = _case(gold, name)
    _, out = _gpu_metrics(*args)
    np.testing.assert_array_equal(out["conf_matrix"], gold[name + ".conf_matrix"])
    for k in SCALARS:
        assert _same(out[k], gold["%s.%s" % (name, k)]), (k, out[k], gold["%s.%s" % (name, k)])


def _random_case(seed, B, L, Ct, H, W, holes):
    rng = np.random.default_rng(seed)
    logits = rng.normal(0, 2, (B, L, H, W)).astype(np.float32)
    logits[rng.uniform(size=logits.shape) < 0.01] = 0.0        # ties and exact zeros
    logits[rng.uniform(size=logits.shape) < 0.01] = 1.0
    cls = rng.integers(0, Ct, (B, H, W))
    seg_full = np.eye(Ct, dtype=np.float32)[cls].transpose(0, 3, 1, 2).copy()
    disp = (rng.uniform(0.1, 8, (B, 1, H, W)) * (rng.uniform(size=(B, 1, H, W)) > (0.25 if holes else -1))).astype(np.float32)
    dp = (disp + rng.normal(0, 2, disp.shape)).astype(np.float32)
    return logits, seg_full, dp, disp


@pytest.mark.gpu
@pytest.mark.parametrize("B,L,Ct,H,W,holes,dtype", [
    (3, 2, 2, 37, 53, False, torch.float32), (2, 19, 20, 33, 65, True, torch.float32), (1, 9, 9, 64, 32, True, torch.float32),
    (2, 2, 2, 48, 40, False, torch.bfloat16), (2, 19, 20, 24, 40, True, torch.bfloat16), (2, 32, 33, 8, 8, False, torch.float32)])
def test_hip_metrics_match_oracle(B, L, Ct, H, W, holes, dtype):
    logits, seg_full, dp, dg = _random_case(B * 100 + L, B, L, Ct, H, W, holes)
    if dtype == torch.bfloat16:     # the oracle sees the values the kernel sees
        rb = lambda a: torch.from_numpy(a).to(torch.bfloat16).float().numpy()
        logits, dp = rb(logits), rb(dp)
    max_disp = 1.0 if not holes else 2.5
    want = MR.step_metrics(logits, seg_full, dp, dg, L, max_disp, holes)
    _, out = _gpu_metrics(logits, seg_full, dp, dg, L, max_disp, holes, dtype)
    np.testing.assert_array_equal(out["conf_matrix"], want["conf_matrix"])
    for k in SCALARS:
        assert _same(out[k], want[k]), (k, out[k], want[k])


@pytest.mark.gpu
def test_hip_metrics_padded_stride_and_partial_calls():
    """Logits inside a padded pixel stride (as the conv kernels emit them); a seg-only and a disp-only call add up
    to the joint call; counters accumulate linearly over repeated calls."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.metrics import StepMetrics
    logits, seg_full, dp, dg = _random_case(5, 2, 2, 2, 40, 56, False)
    dev = torch.device("cuda:0")
    slab, ld = ops.alloc_nhwc(2, 2, 40, 56, torch.float32, dev)
    assert ld > 2
    slab.copy_(torch.from_numpy(logits))
    sf, dpt, dgt = (torch.from_numpy(a).to(dev) for a in (seg_full, dp, dg))
    joint = StepMetrics(2, device=dev)
    joint.update(slab, sf, dpt, dgt)
    parts = StepMetrics(2, device=dev)
    parts.update(slab, sf)
    parts.update(None, sf, dpt, dgt)
    a, b = joint.compute(), parts.compute()
    want = MR.step_metrics(logits, seg_full, dp, dg, 2, 1.0)
    np.testing.assert_array_equal(a["conf_matrix"], want["conf_matrix"])
    for k in SCALARS:
        assert _same(a[k], want[k]) and _same(b[k], a[k]), (k, a[k], b[k], want[k])
    m3, _ = _gpu_metrics(logits, seg_full, dp, dg, 2, 1.0, False, repeat=3)
    m1, _ = _gpu_metrics(logits, seg_full, dp, dg, 2, 1.0, False)
    assert torch.equal(m3.totals()[0], 3 * m1.totals()[0])
    assert torch.allclose(m3.totals()[1], 3 * m1.totals()[1], rtol=1e-12)


@pytest.mark.gpu
def test_hip_metrics_full_size_properties():
    """BASELINE config 2 size (B=8, 256x512): the confusion matrix counts every pixel once, its row sums are the class
    histogram of the target, and TP+FP+FN+TN is the pixel count of image 0."""
    B, L, H, W = 8, 2, 256, 512
    g = torch.Generator(device="cuda").manual_seed(3)
    dev = torch.device("cuda:0")
    logits = torch.randn(B, L, H, W, device=dev, generator=g).to(torch.bfloat16)
    cls = torch.randint(0, L, (B, H, W), device=dev, generator=g)
    seg_full = torch.nn.functional.one_hot(cls, L).permute(0, 3, 1, 2).float().contiguous()
    disp = torch.rand(B, 1, H, W, device=dev, generator=g) * 8 + 0.1
    dp = (disp + torch.randn(B, 1, H, W, device=dev, generator=g)).to(torch.bfloat16)
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.metrics import StepMetrics
    m = StepMetrics(L, max_disp=1.0, device=dev)
    m.update(logits, seg_full, dp, disp)
    out = m.compute()
    conf = out["conf_matrix"]
    assert conf.sum() == B * H * W
    np.testing.assert_array_equal(conf.sum(1), torch.bincount(cls.flatten(), minlength=L).cpu().numpy())
    np.testing.assert_array_equal(conf.sum(0), torch.bincount(logits.float().argmax(1).flatten(), minlength=L).cpu().numpy())
    k = m.totals()[0][L * L:].cpu().numpy()
    assert k[0] + k[1] + k[2] + k[3] == H * W and k[7] == B * H * W
    d = (disp[0, 0] - dp[0, 0].float())
    assert abs(out["dispRMSE"] - float(d.double().pow(2).mean().sqrt())) < 1e-6
    assert out["err"] == float(((dp.float() - disp).abs() > 3).sum())


@pytest.mark.gpu
def test_hip_metrics_reject_bad_arguments():
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import SdhipError
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.metrics import StepMetrics
    dev = torch.device("cuda:0")
    m = StepMetrics(2, device=dev)
    with pytest.raises(SdhipError):
        m.update()
    with pytest.raises(SdhipError):
        m.update(torch.zeros(1, 3, 4, 4, device=dev), torch.zeros(1, 3, 4, 4, device=dev))
    with pytest.raises(SdhipError):
        m.update(torch.zeros(1, 2, 4, 4, device=dev), torch.zeros(1, 4, 4, 4, device=dev))    # Ct not in {L, L+1}
    with pytest.raises(SdhipError):
        m.update(torch.zeros(1, 2, 4, 4), torch.zeros(1, 2, 4, 4))                             # CPU tensors
    with pytest.raises(SdhipError):
        StepMetrics(33, device=dev).update(torch.zeros(1, 33, 4, 4, device=dev), torch.zeros(1, 33, 4, 4, device=dev))


@pytest.mark.gpu
def test_metrics_inside_captured_train_step():
    """StepMetrics.update is one launch with device-resident state, so it rides along in the hipGraph of the training
    step: every replay scores its own outputs (the pixel counts come from the device, not from python)."""
    from oracle import ref_models as R
    from oracle.detweights import fill_state_dict
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N, ops
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.metrics import StepMetrics
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch
    B, H, W = 2, 256, 256
    batch = synthetic_batch(B, H, W)
    model = fill_state_dict(N.minidsnetExt(R.CFG(), labels=2, patch_type='1dcorr'), 5).cuda().train()
    m = StepMetrics(2, max_disp=1.0, device="cuda:0")
    ts = TrainStep(model, dtype=torch.bfloat16, use_graph=True, metrics=m)
    ts(*batch)                       # 2 eager warm-up steps + capture + first replay
    assert ts.graph is not None
    torch.cuda.synchronize()
    assert m.compute()["conf_matrix"].sum() == 3 * B * H * W
    m.reset()
    for _ in range(3):
        ts(*batch)
    torch.cuda.synchronize()
    ops.set_step_context(None)
    out = m.compute()
    k = m.totals()[0][4:].cpu().numpy()
    assert out["conf_matrix"].sum() == 3 * B * H * W
    assert k[0] + k[1] + k[2] + k[3] == 3 * H * W and k[9] == 3 * H * W and k[7] == 3 * B * H * W
    gt_hist = torch.bincount(batch[2].argmax(1).flatten(), minlength=2).cpu().numpy()
    np.testing.assert_array_equal(out["conf_matrix"].sum(1), 3 * gt_hist)
    assert 0.0 <= out["pixelAcc"] <= 1.0 and np.isfinite(out["dispRMSE"]) and out["dispRMSE"] > 0
