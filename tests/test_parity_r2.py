"""Round-2 parity fixtures (all captured from the imported reference by oracle/make_golden.py):
  losses.npz       lossSeg_fn / lossDisp_fn (losses/multiLosses.py) with `cross_entropy lovasz_loss`, roses / garden /
                   cityscapes rules (void class, disp > 0 mask), value + gradients; util/lovasz_losses.py on its own
  dsnetnocorr.npz  dsnet_t2.dsnetnoCorr (= TF baseline_SDnet_small, BASELINE config 1), train + eval
  cfg5.npz         minidsnetExt(aspp=2|0, hanet=1, labels=19) eval with trained-like running statistics + cityscapes loss
  syncbn.npz       sync_batchnorm/_compute_mean_std on split batches vs BatchNorm on the joint batch
CPU tests: the oracle against the fixtures.  GPU tests: the HIP path against the fixtures (f32: 1e-3, the north-star
tolerance), plus the bf16 end-to-end checks of the configuration bench.py times."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_models as R
from oracle.detweights import fill_state_dict, rand_input
from oracle.losses_ref import lovasz_softmax_onehot, train_loss_ref
from test_nets import _check, _net_inputs, train_loss

GDIR = os.path.join(os.path.dirname(__file__), "golden")
LOSS_CASES = ["roses", "garden_absent", "city", "city_allvoid_class", "roses_emptyrow"]


def _loss_case(gold, name):
    L = int(gold[name + ".labels"])
    ds = str(gold[name + ".dataset"])
    logits = torch.from_numpy(gold[name + ".logits"])
    seg_t = torch.from_numpy(gold[name + ".seg_full"][:, :L]).contiguous()       # cityscapes: 20th (void) channel dropped
    return L, ds, logits, seg_t, torch.from_numpy(gold[name + ".disp"]), torch.from_numpy(gold[name + ".disp_pred"])


# ------------------------------------------------------------------ CPU: oracle vs the reference's own loss functions
@pytest.mark.parametrize("name", LOSS_CASES)
def test_oracle_losses_match_reference(name):
    gold = np.load(os.path.join(GDIR, "losses.npz"))
    L, ds, logits, seg_t, disp, disp_pred = _loss_case(gold, name)
    y = logits.clone().requires_grad_(True)
    lv = lovasz_softmax_onehot(y, seg_t, ds == "cityscapes")
    assert abs(float(lv) - float(gold[name + ".lovasz.loss"])) < 1e-5
    assert abs(float(lv) - float(gold[name + ".lovasz_present"])) < 1e-5
    lv.backward()
    assert float((y.grad - torch.from_numpy(gold[name + ".lovasz.grad"])).abs().max()) < 1e-6
    y1, y2, d = (t.clone().requires_grad_(True) for t in (logits, logits, disp_pred))
    tot = train_loss_ref(y1, d, y2, seg_t, disp, True, ds == "cityscapes", ds == "cityscapes")
    want = float(gold[name + ".ce.loss"]) + float(gold[name + ".ce_lovasz.loss"]) + float(gold[name + ".l1.loss"])
    assert abs(float(tot) - want) < 1e-5 * max(1.0, want)
    tot.backward()
    assert float((y1.grad - torch.from_numpy(gold[name + ".ce.grad"])).abs().max()) < 1e-6
    assert float((y2.grad - torch.from_numpy(gold[name + ".ce_lovasz.grad"])).abs().max()) < 1e-6
    assert float((d.grad - torch.from_numpy(gold[name + ".l1.grad"])).abs().max()) < 1e-7


def _dsn_inputs(seed):
    a, b = rand_input(seed, "left", (2, 3, 256, 256)), rand_input(seed, "right", (2, 3, 256, 256))
    seg = F.one_hot((rand_input(seed, "seg", (2, 256, 256)) > 0.5).long(), 2).permute(0, 3, 1, 2).float()
    return a, b, seg, rand_input(seed, "disp", (2, 1, 256, 256), 0.0, 8.0)


def _dsn_loss(outs, seg, disp):
    return torch.mean(torch.sum(-seg * outs[0].float(), 1)) + torch.mean(torch.sum(-seg * outs[2].float(), 1)) + \
        F.l1_loss(outs[1].float(), disp) + F.l1_loss(outs[3].float(), disp)


def test_oracle_dsnetnocorr_matches_golden():
    gold = np.load(os.path.join(GDIR, "dsnetnocorr.npz"))
    a, b, seg, disp = _dsn_inputs(71)
    m = fill_state_dict(R.dsnetnoCorr(R.CFG(), labels=2), 71).train()
    outs = m(a, b)
    for i, name in enumerate(("seg1", "disp", "seg2", "disp2")):
        _check(gold, "dsnetnocorr.train.%s" % name, outs[i], 2e-4)
    assert abs(float(_dsn_loss(outs, seg, disp)) - float(gold["dsnetnocorr.train.loss"])) < 1e-3


def _pos(B, H, W):
    h = (torch.arange(0, H) * 1024 // H).unsqueeze(0).unsqueeze(2).expand(B, -1, W) // 8
    w = (torch.arange(0, W) * 2048 // W).unsqueeze(0).unsqueeze(1).expand(B, H, -1) // 16
    return h, w


def _cfg5_case(gold, tag, ctor, dev):
    kw = dict(aspp=2, hanet=1) if tag.startswith("a2") else dict(aspp=0, hanet=1)
    m = fill_state_dict(ctor(R.CFG(**kw), labels=19, patch_type='1dcorr'), 91)
    sd = m.state_dict()
    pre = tag + ".state."
    for k in gold.files:
        if k.startswith(pre):
            sd[k[len(pre):]].copy_(torch.from_numpy(gold[k]))
    m = m.to(dev).eval()
    a, b = rand_input(91, "left", (2, 3, 256, 256)), rand_input(91, "right", (2, 3, 256, 256))
    cls = (rand_input(91, "cls", (2, 256, 256)) * 20).long().clamp(0, 19)
    seg = F.one_hot(cls, 20).permute(0, 3, 1, 2).float()[:, :19].contiguous()
    disp = rand_input(91, "disp", (2, 1, 256, 256), 0.0, 8.0) * (rand_input(91, "dmask", (2, 1, 256, 256)) > 0.3).float()
    pos = tuple(t.to(dev) for t in _pos(2, 256, 256))
    return m, a.to(dev), b.to(dev), pos, seg.to(dev), disp.to(dev)


@pytest.mark.parametrize("tag", ["a2_hanet_l19", "a0_hanet_l19"])
def test_oracle_cfg5_matches_golden(tag):
    gold = np.load(os.path.join(GDIR, "cfg5.npz"))
    m, a, b, pos, seg, disp = _cfg5_case(gold, tag, R.minidsnetExt, "cpu")
    with torch.no_grad():
        outs = m(a, b, pos)
    for i, name in enumerate(("seg1", "disp", "seg2")):
        _check(gold, "%s.eval.%s" % (tag, name), outs[i], 2e-4)
    loss = train_loss_ref(outs[0], outs[1], outs[2], seg, disp, True, True, True)
    assert abs(float(loss) - float(gold[tag + ".eval.loss"])) < 1e-3


@pytest.mark.parametrize("name", ["two_ranks", "four_ranks_tiny"])
def test_sync_bn_sums_reproduce_reference_statistics(name):
    """What the ranks exchange — per-shard (sum, sum of squares, count) — turned into mean / invstd / running statistics by
    parallel.bn_scale_shift_from_sums equals the reference's `_compute_mean_std` on the same shards
    (sync_batchnorm/batchnorm.py:114-126) and BatchNorm on the joint batch.  The vendored module clamps the variance at eps
    (`bias_var.clamp(eps) ** -0.5`) where the live path (nn.SyncBatchNorm, torch_implementation.py:739) adds eps: the
    live rule is what is implemented; against the clamp rule invstd agrees to eps / var."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import parallel
    g = np.load(os.path.join(GDIR, "syncbn.npz"))
    x = torch.from_numpy(g[name + ".x"]).double()
    C = x.shape[1]
    shards = x.chunk(int(g[name + ".parts"]), 0)
    s1 = sum(s.sum((0, 2, 3)) for s in shards)
    s2 = sum((s * s).sum((0, 2, 3)) for s in shards)
    n = sum(s.numel() // C for s in shards)
    scale, shift, mean, var = parallel.bn_scale_shift_from_sums(s1, s2, n, torch.ones(C), torch.zeros(C), 1e-5)
    np.testing.assert_allclose(mean.numpy(), g[name + ".mean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(scale.numpy(), g[name + ".inv_std_clamp"], rtol=2e-5 + 1e-5 / float(var.min()))
    y = x * scale.view(1, C, 1, 1) + shift.view(1, C, 1, 1)
    np.testing.assert_allclose(y.numpy(), g[name + ".y_joint"], rtol=1e-4, atol=1e-5)
    rm = 0.9 * g[name + ".rm0"] + 0.1 * mean.numpy()
    rv = 0.9 * g[name + ".rv0"] + 0.1 * (var.numpy() * n / (n - 1))
    for want in (g[name + ".rm_sync"], g[name + ".rm_joint"]):
        np.testing.assert_allclose(rm, want, rtol=1e-5, atol=1e-6)
    for want in (g[name + ".rv_sync"], g[name + ".rv_joint"]):
        np.testing.assert_allclose(rv, want, rtol=1e-4, atol=1e-6)


# ------------------------------------------------------------------ GPU: HIP path vs the fixtures
@pytest.mark.gpu
@pytest.mark.parametrize("name", LOSS_CASES)
def test_hip_losses_match_reference(name):
    """sdhip_ce_loss + sdhip_lovasz_softmax + sdhip_l1_loss (ops.train_loss) against the reference's lossSeg_fn /
    lossDisp_fn: value 1e-5, gradients 1e-6 (f32)."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    gold = np.load(os.path.join(GDIR, "losses.npz"))
    L, ds, logits, seg_t, disp, disp_pred = _loss_case(gold, name)
    y1, y2, d = (t.cuda().requires_grad_(True) for t in (logits, logits, disp_pred))
    tot = ops.train_loss(y1, d, y2, seg_t.cuda(), disp.cuda(), True, ds == "cityscapes", ds == "cityscapes")
    tot.backward()
    want = float(gold[name + ".ce.loss"]) + float(gold[name + ".ce_lovasz.loss"]) + float(gold[name + ".l1.loss"])
    assert abs(float(tot) - want) < 1e-5 * max(1.0, want), (float(tot), want)
    assert float((y1.grad.cpu() - torch.from_numpy(gold[name + ".ce.grad"])).abs().max()) < 1e-6
    assert float((y2.grad.cpu() - torch.from_numpy(gold[name + ".ce_lovasz.grad"])).abs().max()) < 1e-6
    assert float((d.grad.cpu() - torch.from_numpy(gold[name + ".l1.grad"])).abs().max()) < 1e-7
    # without the Lovasz term the second head carries the plain CE gradient
    y1, y2, d = (t.cuda().requires_grad_(True) for t in (logits, logits, disp_pred))
    ops.train_loss(y1, d, y2, seg_t.cuda(), disp.cuda(), False, ds == "cityscapes", ds == "cityscapes").backward()
    assert float((y2.grad.cpu() - torch.from_numpy(gold[name + ".ce.grad"])).abs().max()) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_hip_dsnetnocorr_matches_golden(mode):
    """dsnetnoCorr = PyTorch port of the TF baseline_SDnet_small graph (BASELINE config 1; models/dsnet_t2.py:620-823)."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N
    gold = np.load(os.path.join(GDIR, "dsnetnocorr.npz"))
    m = fill_state_dict(N.dsnetnoCorr(R.CFG(), labels=2), 71).cuda()
    m.train() if mode == "train" else m.eval()
    a, b, seg, disp = (t.cuda() for t in _dsn_inputs(71))
    outs = m(a, b)
    loss = _dsn_loss(outs, seg, disp)
    loss.backward()
    p = "dsnetnocorr.%s" % mode
    for i, name in enumerate(("seg1", "disp", "seg2", "disp2")):
        _check(gold, "%s.%s" % (p, name), outs[i], 1e-3)
    want = float(gold[p + ".loss"])
    assert abs(loss.item() - want) <= 1e-3 * max(1.0, abs(want))
    if mode == "train":
        acc = {}
        for k, q in m.named_parameters():
            if q.grad is not None:
                top = k.split(".")[0]
                acc[top] = acc.get(top, 0.0) + float(q.grad.double().pow(2).sum())
        for top, v in acc.items():
            key = "%s.gnorm.%s" % (p, top)
            if key in gold.files:
                w = float(gold[key])
                assert abs(np.sqrt(v) - w) <= 3e-2 * max(w, 1e-3), (key, np.sqrt(v), w)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a2_hanet_l19", "a0_hanet_l19"])
def test_hip_cfg5_matches_golden(tag):
    """BASELINE config 5's network at 256x256: outputs 1e-3, the cityscapes loss (19 classes + void, disp > 0 mask) 1e-3,
    gradient norms per submodule through the eval-mode network.  With aspp=2 the HANet head is built but never applied
    (as upstream, models/dsnet_t2.py:1287-1289): its parameters receive no gradient in either implementation."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N, ops
    gold = np.load(os.path.join(GDIR, "cfg5.npz"))
    m, a, b, pos, seg, disp = _cfg5_case(gold, tag, N.minidsnetExt, "cuda")
    outs = m(a, b, pos)
    loss = ops.train_loss(outs[0], outs[1], outs[2], seg, disp, True, True, True)
    loss.backward()
    for i, name in enumerate(("seg1", "disp", "seg2")):
        _check(gold, "%s.eval.%s" % (tag, name), outs[i], 1e-3)
    want = float(gold[tag + ".eval.loss"])
    assert abs(float(loss) - want) <= 1e-3 * max(1.0, want), (float(loss), want)
    acc = {}
    for k, q in m.named_parameters():
        if q.grad is not None:
            top = k.split(".")[0]
            acc[top] = acc.get(top, 0.0) + float(q.grad.double().pow(2).sum())
    checked = 0
    for top, v in acc.items():
        key = "%s.eval.gnorm.%s" % (tag, top)
        if key in gold.files:
            w = float(gold[key])
            assert abs(np.sqrt(v) - w) <= 3e-2 * max(w, 1e-3), (key, np.sqrt(v), w)
            checked += 1
    assert checked >= 10
    if tag.startswith("a2"):
        assert "hanet_last" not in acc or acc["hanet_last"] == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,aspp", [(2, 512, 1024, 2), (2, 512, 1024, 0), (4, 512, 960, 0)])
def test_hip_large_config_properties(B, H, W, aspp):
    """BASELINE configs 4 / 5 at their stated image sizes (960x512, 1024x512; bf16 train step, 19 classes + HANet):
    shapes, finiteness, a loss that falls over 3 steps, and metric counters that add up to the pixel count."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N, ops
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.metrics import StepMetrics
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch
    torch.manual_seed(0)
    m = N.minidsnetExt(N.CFG(aspp=aspp, hanet=1), labels=19, patch_type='1dcorr').cuda().train()
    left, right, seg, disp = synthetic_batch(B, H, W, labels=19, seed=7)
    pos = tuple(t.cuda() for t in _pos(B, H, W))
    outs = m(left.bfloat16(), right.bfloat16(), pos)
    assert tuple(outs[0].shape) == (B, 19, H, W) and tuple(outs[1].shape) == (B, 1, H, W) and tuple(outs[2].shape) == (B, 19, H, W)
    for o in outs[:3]:
        assert torch.isfinite(o.float()).all()
    loss = ops.train_loss(outs[0], outs[1], outs[2], seg, disp, True, True, True)
    loss.backward()
    assert torch.isfinite(loss).all()
    bad = [k for k, p in m.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    g = [p.grad for p in m.parameters() if p.grad is not None]
    assert len(g) > 500 and not bad, (len(g), bad[:8])
    sm = StepMetrics(19, device="cuda")
    sm.update(outs[2].detach(), seg, outs[1].detach(), disp)
    res = sm.compute()
    assert int(np.asarray(res["conf_matrix"]).sum()) == B * H * W
    del outs, loss, g
    torch.cuda.empty_cache()


# ------------------------------------------------------------------ GPU: bf16 end to end
def _gnorms(model, depth=1):
    acc = {}
    for k, q in model.named_parameters():
        if q.grad is not None:
            top = ".".join(k.split(".")[:depth])
            acc[top] = acc.get(top, 0.0) + float(q.grad.double().pow(2).sum())
    return {k: np.sqrt(v) for k, v in acc.items()}


def _rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12))


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W", [(2, 256, 256), (8, 256, 512)])
def test_hip_minidsnet_bf16_error_is_bounded_by_the_networks_own_sensitivity(B, H, W):
    """bf16 end to end against the f32 path — with a bound derived from a measurement, because a fixed one does not exist:
    at the fixture's random weights the train-mode network (120 batch-statistics BatchNorm layers in sequence) is chaotic.
    The f32 path's OWN response to a 1e-4 relative perturbation of the input images is 1.7 % (seg1), 0.4 % (disp) and 11 %
    (seg2) relative L2 — amplification factors 170 / 40 / 1100 (tests/diag/gpu_r2_diag.py).  bf16 storage perturbs every
    stored activation by <= 2^-9 = 20 x 1e-4; ~150 stored tensors in sequence add up in quadrature (x 12).  The test
    measures the f32 response r to the 1e-4 perturbation and requires  err_bf16 <= min(0.2, 50 * r)  on the disparity head,
    i.e. bf16 behaves like a perturbation of at most ~2 bf16 ulps per tensor (measured: 20-32 x r; a wrong kernel gives O(1)
    errors there); the segmentation heads get a direction check instead of a cap above 1 (see below).  The DenseNet taps, where amplification is still small, are bounded directly:
    0.3 / 1.8 / 3.8 / 9 / 14.5 % measured -> 1 / 4 / 8 / 18 / 25 %.  Loss within 2 %.  (8, 256, 512) is the bench shape."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import synthetic_batch
    left, right, seg, disp = synthetic_batch(B, H, W, seed=3)
    mk = lambda: fill_state_dict(N.minidsnetExt(R.CFG(aspp=0), labels=2, patch_type='1dcorr'), 31).cuda().train()
    with torch.no_grad():
        m = mk()
        o32 = m(left, right)
        t32 = mk().resnet_features(torch.cat([left, right]), groups=2)
        g = torch.Generator(device="cuda").manual_seed(11)
        noise = lambda t: t * (1 + 1e-4 * torch.randn(t.shape, device="cuda", generator=g))
        o32p = mk()(noise(left), noise(right))
        o16 = mk()(left.bfloat16(), right.bfloat16())
        t16 = mk().resnet_features(torch.cat([left, right]).bfloat16(), groups=2)
    for i, cap in enumerate((0.01, 0.04, 0.08, 0.18, 0.25)):
        assert _rel(t16[i], t32[i]) <= cap, ("tap", i, _rel(t16[i], t32[i]))
    for i, name in enumerate(("seg1", "disp", "seg2")):
        r = _rel(o32p[i], o32[i])
        e = _rel(o16[i], o32[i])
        if name == "disp":       # the well-conditioned head: a fixed cap far below what a zero output scores (1.0)
            assert e <= min(0.2, 50.0 * r), (name, e, r)
            continue
        # Segmentation heads: 50 * r exceeds 1 at these weights (measured e = 0.33-0.36 / 0.87-0.89), and a relative-L2 cap
        # above 1 cannot fail — so none is asserted.  What is asserted is what a zero, constant or unrelated output fails:
        # the bf16 logits still point the f32 way (cosine > 0.3; measured 0.93 / 0.6).  The same kernels are bounded per head
        # to a few per cent where the network is not chaotic: tests/test_bf16_parity.py (eval mode).
        a16, a32 = o16[i].float().flatten(), o32[i].float().flatten()
        cos = float(torch.dot(a16, a32) / (a16.norm() * a32.norm()).clamp_min(1e-20))
        assert cos > 0.3 and torch.isfinite(a16).all(), (name, cos, e, r)
    l32 = float(train_loss(o32, seg, disp)); l16 = float(train_loss(o16, seg, disp))
    assert abs(l16 - l32) <= 2e-2 * l32, (l16, l32)


@pytest.mark.gpu
def test_hip_bench_configuration_bf16_graph_vs_cpu_oracle():
    """EXACTLY what bench.py times — minidsnetExt, B = 8, 256x512, bf16, one hipGraph replay of forward + loss + backward
    + Adam — against the f32 CPU oracle on the same batch and weights: the loss of the replayed step within 2 %, replays
    reproducible, and the disparity head (the well-conditioned one: see the sensitivity test above) within 20 % relative L2."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N, ops
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch
    B, H, W = 8, 256, 512
    left, right, seg, disp = synthetic_batch(B, H, W, device="cpu")
    torch.set_num_threads(16)
    ref = fill_state_dict(R.minidsnetExt(R.CFG(aspp=0), labels=2, patch_type='1dcorr'), 31).train()
    with torch.no_grad():
        ro = ref(left, right)
        want = float(train_loss_ref(ro[0], ro[1], ro[2], seg, disp, True))
    m = fill_state_dict(N.minidsnetExt(R.CFG(aspp=0), labels=2, patch_type='1dcorr'), 31).cuda().train()
    ts = TrainStep(m, dtype=torch.bfloat16, use_graph=True, lr=0.0)       # lr 0: every step sees the fixture's weights
    batch = [t.cuda() for t in (left, right, seg, disp)]
    got = float(ts(*batch))              # 2 eager warm-up steps, capture, first replay
    junk = [torch.randn(1000, device="cuda") for _ in range(500)]      # unrelated allocations between replays must not matter
    got2 = float(ts(*batch))             # a second replay
    assert ts.use_graph and ts.graph is not None
    assert abs(got - want) <= 2e-2 * max(1.0, abs(want)), (got, want)
    assert abs(got2 - got) <= 1e-3 * max(1.0, abs(got)), (got, got2)
    ops.set_step_context(None)
    with torch.no_grad():
        outs = m(batch[0].bfloat16(), batch[1].bfloat16())
    assert _rel(outs[1].cpu(), ro[1]) <= 0.2, _rel(outs[1].cpu(), ro[1])
    del junk


@pytest.mark.gpu
def test_bf16_trains_like_f32():
    """50 optimizer steps on one batch, f32 path vs bf16 path from the same weights (shipped Adam settings, lr 1.5e-3):
    the loss must fall in both and the two trajectories must stay together (bf16 within 10 % of f32 averaged over the last
    10 steps)."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N, ops
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch
    batch = synthetic_batch(2, 256, 256, seed=5)
    traj = {}
    for dtype in (torch.float32, torch.bfloat16):
        m = fill_state_dict(N.minidsnetExt(R.CFG(aspp=0), labels=2, patch_type='1dcorr'), 31).cuda().train()
        ts = TrainStep(m, dtype=dtype, use_graph=True)
        traj[dtype] = [float(ts(*batch)) for _ in range(48)]     # + 2 warm-up steps inside the first call
        ops.set_step_context(None)
        del ts, m
    f, h = np.array(traj[torch.float32]), np.array(traj[torch.bfloat16])
    assert np.isfinite(f).all() and np.isfinite(h).all()
    # (random labels: what can be learnt in 50 steps is the class prior and the disparity scale — about 20 % of the loss)
    assert f[-10:].mean() < 0.9 * f[:3].mean() and h[-10:].mean() < 0.9 * h[:3].mean(), (f[:3], f[-10:], h[:3], h[-10:])
    assert abs(h[-10:].mean() - f[-10:].mean()) <= 0.05 * f[-10:].mean(), (f[-10:].mean(), h[-10:].mean())
    assert np.abs(h - f).max() <= 0.1 * f.max(), (f, h)
