"""Training step: eager vs hipGraph replay, direct-gradient / arena services vs plain autograd, Adam vs torch.optim.Adam."""
import copy

import pytest
import torch

from oracle import ref_models as R
from oracle.detweights import fill_state_dict
from oracle.losses_ref import lovasz_softmax_onehot


def _model():
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N
    torch.manual_seed(0)
    return fill_state_dict(N.minidsnetExt(R.CFG(), labels=2, patch_type='1dcorr'), 5).cuda().train()


@pytest.mark.gpu
def test_step_services_match_plain_autograd():
    """Gradients accumulated directly into the flat buffer (arena workspaces, batched weight packing) equal the
    gradients autograd returns without those services."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch
    batch = synthetic_batch(2, 256, 256)
    m0 = _model()
    ops.set_step_context(None)
    outs = m0(batch[0], batch[1])
    ops.train_loss(outs[0], outs[1], outs[2], batch[2], batch[3], True).backward()
    want = {k: p.grad.clone() for k, p in m0.named_parameters() if p.grad is not None}
    m1 = _model()
    ts = TrainStep(m1, dtype=torch.float32, use_graph=False, lr=0.0)     # lr 0: parameters stay put
    ts(*batch)            # measuring step (plain autograd path)
    ts(*batch)            # armed step: arena + frozen packs + direct gradients
    ops.set_step_context(None)
    for k, p in m1.named_parameters():
        if k in want:
            ref = want[k]
            err = float(torch.linalg.norm(p.grad - ref) / torch.linalg.norm(ref).clamp_min(1e-12))
            assert err < 1e-3, (k, err)
    n5 = m1.resnet_features.resnet_features.norm5
    assert int(n5.num_batches_tracked) == 4   # 2 steps x 2 statistics groups


@pytest.mark.gpu
def test_graph_replay_matches_eager():
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch
    batch = synthetic_batch(2, 256, 256)
    losses = {}
    for graph in (False, True):
        ts = TrainStep(_model(), dtype=torch.float32, use_graph=graph, lr=1e-4)
        seq = []
        if graph:
            ts.capture(*batch, warmup=2)       # 2 eager warm-up steps, then the captured step is replayed
            for _ in range(2):
                seq.append(float(ts(*batch)))
        else:
            for _ in range(5):
                seq.append(float(ts(*batch)))
            seq = seq[2:4]                     # steps 3 and 4 (capture only records; the first replay is step 3)
        losses[graph] = seq
        ops.set_step_context(None)
    # first replayed step: same state as the eager run; later steps drift with the f32-atomic summation order
    assert abs(losses[False][0] - losses[True][0]) <= 2e-3 * max(1.0, abs(losses[False][0])), losses
    assert abs(losses[False][1] - losses[True][1]) <= 2e-2 * max(1.0, abs(losses[False][1])), losses


@pytest.mark.gpu
def test_adam_kernel_matches_torch():
    from pmt_learning_for_semantic_segmentation_and_disparity_amd._lib import call, ptr, stream_ptr
    torch.manual_seed(1)
    n = 1003
    p = torch.randn(n, device="cuda"); g = [torch.randn(n, device="cuda") for _ in range(3)]
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=0.0015, eps=1e-7)
    m = torch.zeros(n + 1, device="cuda")[:n]; v = torch.zeros_like(p); bp = torch.ones(2, device="cuda")
    m = torch.zeros(n, device="cuda")
    for gi in g:
        ref.grad = gi.clone(); opt.step()
        call("sdhip_adam_step", ptr(p), ptr(gi), ptr(m), ptr(v), ptr(bp), n, 0.0015, 0.9, 0.999, 1e-7, 0.0, 1.0, stream_ptr())
    assert float((p - ref.detach()).abs().max()) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 2e-2)])
def test_train_loss_matches_reference_formulas(dtype, tol):
    """CE + CE + Lovasz + L1 value and gradients vs the plain-torch restatement of the reference formulas
    (util/utilTorchLoss.py:373-378, util/lovasz_losses.py:153-199, losses/multiLosses.py:70-72,141)."""
    import torch.nn.functional as F
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    torch.manual_seed(3)
    B, C, H, W = 2, 2, 64, 96
    seg1 = torch.randn(B, C, H, W); seg2 = torch.randn(B, C, H, W); disp = torch.rand(B, 1, H, W) * 8
    lab = torch.randint(0, C, (B, H, W))
    seg_t = F.one_hot(lab, C).permute(0, 3, 1, 2).float().contiguous(); disp_t = torch.rand(B, 1, H, W) * 8
    if dtype == torch.bfloat16:
        seg1, seg2, disp = (t.bfloat16().float() for t in (seg1, seg2, disp))
    r = [t.clone().requires_grad_(True) for t in (seg1, disp, seg2)]
    ce = lambda y: torch.mean(torch.sum(-seg_t * F.log_softmax(y, 1), 1))
    want = ce(r[0]) + ce(r[2]) + lovasz_softmax_onehot(r[2], seg_t) + F.l1_loss(r[1], disp_t)
    want.backward()
    g = [t.cuda().to(dtype).requires_grad_(True) for t in (seg1, disp, seg2)]
    got = ops.train_loss(g[0], g[1], g[2], seg_t.cuda(), disp_t.cuda(), True)
    got.backward()
    assert abs(float(got) - float(want)) <= tol * max(1.0, abs(float(want)))
    for a, b in zip(g, r):
        scale = float(b.grad.abs().max())
        assert float((a.grad.float().cpu() - b.grad).abs().max()) <= max(tol, 2e-3) * scale


@pytest.mark.gpu
def test_lovasz_single_class_present():
    """classes='present': a class with no pixel is skipped (and the mean runs over the present ones only)."""
    import torch.nn.functional as F
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    torch.manual_seed(4)
    B, C, H, W = 1, 3, 32, 32
    seg = torch.randn(B, C, H, W)
    lab = torch.zeros(B, H, W, dtype=torch.long); lab[:, :, 16:] = 2          # class 1 absent
    seg_t = F.one_hot(lab, C).permute(0, 3, 1, 2).float().contiguous()
    zero = torch.zeros(B, 1, H, W)
    r = seg.clone().requires_grad_(True)
    ce = lambda y: torch.mean(torch.sum(-seg_t * F.log_softmax(y, 1), 1))
    want = 2 * ce(r) + lovasz_softmax_onehot(r, seg_t)
    want.backward()
    g1 = seg.cuda().requires_grad_(True); g2 = seg.cuda().requires_grad_(True)
    got = ops.train_loss(g1, zero.cuda().requires_grad_(True), g2, seg_t.cuda(), zero.cuda(), True)
    got.backward()
    assert abs(float(got) - float(want)) < 1e-4
    assert float(((g1.grad + g2.grad).cpu() - r.grad).abs().max()) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,H,W,void", [(1, 19, 50, 83, True), (2, 5, 64, 100, False), (8, 2, 256, 512, False), (2, 19, 256, 512, True)])
def test_lovasz_segmented_sort_sizes(B, C, H, W, void):
    """The hand-written segmented radix sort behind sdhip_lovasz_softmax (lovasz.hip: a wave owns 2048 elements, four stable
    8-bit passes, every class in the same launches): sizes that are no multiple of a wave's share, 19 classes with void
    pixels, and the benchmark's full size, against the restated reference formula run on the same device (torch.sort);
    two calls agree bit for bit (nothing in the sort depends on timing)."""
    import torch.nn.functional as F
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + C)
    seg = torch.randn(B, C, H, W, generator=g) * 2
    lab = torch.randint(0, C + (1 if void else 0), (B, H, W), generator=g)
    seg_t = F.one_hot(lab, C + (1 if void else 0)).permute(0, 3, 1, 2).float()[:, :C].contiguous()   # label C = void: an all-zero row
    zero = torch.zeros(B, 1, H, W)
    from oracle.losses_ref import train_loss_ref
    st = seg_t.cuda()
    r1 = seg.clone().cuda().requires_grad_(True); r = seg.clone().cuda().requires_grad_(True)
    want = train_loss_ref(r1, zero.cuda(), r, st, zero.cuda(), True, void, void)
    want.backward()
    outs = []
    for _ in range(2):
        g1 = seg.cuda().requires_grad_(True); g2 = seg.cuda().requires_grad_(True)
        got = ops.train_loss(g1, zero.cuda().requires_grad_(True), g2, st, zero.cuda(), True, void, void)
        got.backward()
        outs.append((got.detach().clone(), g2.grad.clone(), g1.grad.clone()))
    assert torch.equal(outs[0][1], outs[1][1])
    assert abs(float(outs[0][0]) - float(want)) < 2e-5 * max(1.0, abs(float(want))), (float(outs[0][0]), float(want))
    assert float((outs[0][1] - r.grad).abs().max()) < 1e-7 + 1e-4 * float(r.grad.abs().max())


def _aspp_model():
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N
    torch.manual_seed(0)
    return fill_state_dict(N.minidsnetExt(R.CFG(aspp=1), labels=2, patch_type='1dcorr'), 5).cuda().train()


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [False, True])
def test_dropout_seed_advances_every_step(graph):
    """ASPP's Dropout(0.5) (models/aspp.py:79,95) must draw a new mask each training step, eager and under graph replay
    (the add on the device-resident seed is part of the captured step); forward and backward of one step share the mask."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch
    batch = synthetic_batch(2, 256, 256)
    ts = TrainStep(_aspp_model(), dtype=torch.float32, use_graph=graph, lr=0.0)   # lr 0: only the masks change the loss
    seed = ts.ctx.seed                       # the step's own device-resident seed
    other = TrainStep(_aspp_model(), dtype=torch.float32, use_graph=False, lr=0.0)   # a second step must not restart the first one's stream
    seeds, losses = [], []
    for _ in range(5):
        losses.append(float(ts(*batch)))
        seeds.append(int(seed.item()))
    ops.set_step_context(None)
    assert [b - a for a, b in zip(seeds, seeds[1:])] == [1, 1, 1, 1], seeds
    assert ts.steps_done == (7 if graph else 5)      # the first graph call runs 2 eager warm-up steps before its replay
    assert len({round(l, 6) for l in losses[2:]}) == 3, losses     # same weights, same batch, different masks


@pytest.mark.gpu
def test_dropout_mask_is_shared_by_forward_and_backward():
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    x = torch.randn(2, 8, 16, 16, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ops.rng_reseed("cuda", rank=0)
    y = ops.dropout(x, 0.5, True, 7)
    y.sum().backward()
    kept = (y != 0)
    assert torch.equal(x.grad != 0, kept) and 0.35 < float(kept.float().mean()) < 0.65
    ops.rng_reseed("cuda", rank=1)
    assert not torch.equal(ops.dropout(x, 0.5, True, 7) != 0, kept)     # other ranks draw other masks


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["python_raise", "hip_invalidate"])
def test_failed_capture_falls_back_to_a_consistent_eager_step(how, capfd):
    """A hipGraph capture that raises leaves TrainStep usable: the steps that follow equal those of an eager-only run
    (the recording pass executed nothing, so parameters / moments / running statistics are those of the warm-up)."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch
    batch = synthetic_batch(2, 256, 256)
    ref = TrainStep(_model(), dtype=torch.float32, use_graph=False, lr=1e-4)
    want = [float(ref(*batch)) for _ in range(4)]
    ops.set_step_context(None)
    ts = TrainStep(_model(), dtype=torch.float32, use_graph=True, lr=1e-4)

    def boom():
        if how == "hip_invalidate":
            torch.cuda.synchronize()      # illegal while capturing: HIP invalidates the capture (what an uncapturable call does)
        raise RuntimeError("injected capture failure")
    ts._capture_fault = boom
    got = [float(ts(*batch))]            # 2 eager warm-up steps + failed capture + 1 eager step = step 3
    assert ts.use_graph is False and ts.graph is None and ts.steps_done == 3
    got.append(float(ts(*batch)))
    ops.set_step_context(None)
    assert abs(got[0] - want[2]) <= 2e-3 * max(1.0, abs(want[2])), (got, want)
    assert abs(got[1] - want[3]) <= 2e-2 * max(1.0, abs(want[3])), (got, want)
    n5 = ts.model.resnet_features.resnet_features.norm5
    assert int(n5.num_batches_tracked) == 8      # 4 executed steps x 2 statistics groups; the recording pass counted nothing
    err = capfd.readouterr().err
    assert "continuing WITHOUT a graph" in err and "could not" not in err, err
    torch.randn(8, device="cuda")                # torch's own CUDA generator is usable again (it was in capture mode)


@pytest.mark.gpu
def test_failed_capture_keeps_torchs_cuda_random_stream():
    """After a failed capture torch's default CUDA generator continues where it was before the capture began (it
    used to be re-seeded at offset 0, so later draws replayed the stream from its start)."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch
    batch = synthetic_batch(2, 256, 256)
    ts = TrainStep(_model(), dtype=torch.float32, use_graph=True, lr=1e-4)
    torch.manual_seed(77)
    first = torch.rand(1000, device="cuda")      # consumes part of the stream: the state at capture time is NOT the start
    want_next = None
    st = torch.cuda.get_rng_state()
    want_next = torch.rand(1000, device="cuda")
    torch.cuda.set_rng_state(st)

    def boom():
        raise RuntimeError("injected capture failure")
    ts._capture_fault = boom
    ts(*batch)
    ops.set_step_context(None)
    got = torch.rand(1000, device="cuda")
    assert torch.equal(got, want_next) and not torch.equal(got, first)


@pytest.mark.gpu
def test_captured_step_consists_of_kernel_nodes_only():
    """No memset node inside the captured training step: hipMemsetAsync nodes were observed to stop clearing after unrelated
    allocations between replays on ROCm 7.2 (sdhip_common.h), so every clear is a kernel — the Lovasz counters, DenseNet's
    incoming-statistics replicas (sdhip_channel_stats), odd-sized bf16 buffers included."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops, _lib
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch
    batch = synthetic_batch(2, 256, 256)
    ts = TrainStep(_model(), dtype=torch.bfloat16, use_graph=True, lr=1e-4)
    ts.debug_graph = True
    l1 = float(ts(*batch))
    l2 = float(ts(*batch))                       # a kept graph still replays
    ops.set_step_context(None)
    assert ts.graph is not None and l2 == l2 and l1 == l1
    nodes = _lib.graph_node_counts(ts.graph)
    assert nodes["kernel"] > 500 and nodes["memset"] == 0, nodes


@pytest.mark.gpu
def test_zero_kernel_any_alignment():
    """sdhip_zero_async clears buffers of any alignment and length with a kernel (an odd bf16 element count used to fall back
    to hipMemsetAsync): exercised through the HANet row-pool backward, which clears its gradient map first."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    for C, H, W in ((3, 5, 7), (1, 9, 3), (8, 4, 4)):
        x = torch.randn(1, C, H, W, device="cuda").bfloat16().requires_grad_(True)
        y = ops.rowpool_max(x, 2)
        y.float().sum().backward()
        g = x.grad.float()
        assert torch.isfinite(g).all() and float(g.sum()) == float(C * 2)     # one winner per (channel, output row)


@pytest.mark.gpu
def test_conv2downup_dropout_training():
    """Conv2DownUp with Dropout(p > 0) in training mode (models/dsnet_t2.py:85-93): runs, drops about p of the activations
    after the ReLU and before the skip add, is the identity path in eval mode."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N, ops
    torch.manual_seed(0)
    blk = N.Conv2DownUp(16, 16, 3, lastLayer=True, dropout=0.5).cuda().train()
    x = torch.randn(2, 16, 32, 32, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ops.rng_reseed("cuda")
    y = blk(x)
    frac0 = float((y == 0).float().mean())
    assert 0.65 < frac0 < 0.85, frac0          # relu zeros (~half) plus dropped survivors (half of the rest)
    y.float().pow(2).mean().backward()
    assert torch.isfinite(x.grad).all() and float(x.grad.abs().sum()) > 0
    blk.eval()
    ye = blk(x)
    assert 0.3 < float((ye == 0).float().mean()) < 0.7


@pytest.mark.gpu
def test_graph_with_side_stream_matches_eager():
    """The data-parallel configuration of the capture on one GPU: weight gradients on a second captured stream (forked and
    joined inside the graph, as TrainStep does when world_size > 1), warm-up and capture on one stream.  Replays must equal
    the eager steps of the same state."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch
    batch = synthetic_batch(2, 256, 256)
    ref = TrainStep(_model(), dtype=torch.float32, use_graph=False, lr=1e-4)
    want = [float(ref(*batch)) for _ in range(4)]
    ops.set_step_context(None)
    ts = TrainStep(_model(), dtype=torch.float32, use_graph=True, lr=1e-4, use_side_stream=True)
    got = [float(ts(*batch)) for _ in range(2)]          # steps 3 and 4 (two warm-up steps inside the first call)
    ops.set_step_context(None)
    assert ts.use_graph and ts.graph is not None and ts.ctx.side is not None
    assert abs(got[0] - want[2]) <= 2e-3 * max(1.0, abs(want[2])), (got, want)
    assert abs(got[1] - want[3]) <= 2e-2 * max(1.0, abs(want[3])), (got, want)


@pytest.mark.gpu
def test_pack_table_holds_only_the_steps_own_weights():
    """The batched weight-pack table a TrainStep replays every step (inside its hipGraph) must reference ONLY its own model's
    parameters: with a second model alive (its packs cached too) and a per-step temporary weight in the cache, every source
    address of the table lies inside the step's flat parameter buffer."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch
    batch = synthetic_batch(2, 256, 256)
    other = _model()                                     # another live model whose weights get packed and cached
    ops.set_step_context(None)
    with torch.no_grad():
        other(batch[0], batch[1])
    ts = TrainStep(_model(), dtype=torch.float32, use_graph=False, lr=1e-4)
    for _ in range(2):
        ts(*batch)                                       # measuring step + the step that arms the services
    assert ts.pack_desc is not None and ts.pack_desc.shape[0] > 100
    lo = ts.flat_p.data_ptr()
    hi = lo + ts.flat_p.numel() * ts.flat_p.element_size()
    src = ts.pack_desc[:, 0].cpu()
    assert bool(((src >= lo) & (src < hi)).all())
    foreign = {p.data_ptr() for p in other.parameters()}
    assert not (set(src.tolist()) & foreign)
    del other
    ops.set_step_context(None)


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [False, True])
def test_gradient_accumulation_matches_reference_rule(graph):
    """`-acmt_grad K` (torch_implementation.py:335,362,390-397): the loss of each of K consecutive batches is divided by K and
    back-propagated, the optimizer steps once.  TrainStep(accumulate=3): three calls (one per micro-batch) produce the
    parameters of ONE Adam step on the mean of the three micro-batch gradients — checked against three single steps' own
    gradients (lr 0 runs on the same weights) fed to torch.optim.Adam, eagerly and with one hipGraph per kind of call
    (opening / middle / closing call of a cycle)."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch
    batches = [synthetic_batch(2, 256, 256, seed=100 + i) for i in range(3)]
    # gradient of every micro-batch at the initial weights (eval-mode BatchNorm would hide nothing here: train mode, but the
    # running statistics do not enter the train-mode forward, so the three gradients are independent of the call order)
    ref = TrainStep(_model(), dtype=torch.float32, use_graph=False, lr=0.0)
    gsum = torch.zeros_like(ref.flat_g)
    for b in batches:
        ref(*b)
        gsum += ref.flat_g
    w0 = ref.flat_p.clone()
    ops.set_step_context(None)
    p = w0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p], lr=1e-3, betas=(0.9, 0.999), eps=1e-7)
    p.grad = gsum / 3
    opt.step()
    if not graph:
        ts = TrainStep(_model(), dtype=torch.float32, use_graph=False, lr=1e-3, accumulate=3)
        assert torch.equal(ts.flat_p, w0)
        for b in batches:
            ts(*b)
        ops.set_step_context(None)
        assert ts.steps_done == 1
        d_want, d_got = (p.detach() - w0), (ts.flat_p - w0)
        assert float(d_want.norm()) > 0
        # Adam's first step moves every parameter by ~lr * sign(g): compare the update vectors, and the accumulated gradient
        assert float((d_got - d_want).norm() / d_want.norm()) < 2e-2
        assert float((ts.flat_g / 3 - gsum / 3).norm() / (gsum / 3).norm()) < 2e-2
        return
    # graph mode (one hipGraph per kind of call: opening / middle / closing call of a cycle).  lr 0 keeps the weights at w0, so
    # the buffer the closing call hands to Adam can be compared directly: the sum of the three micro-batch gradients
    ts = TrainStep(_model(), dtype=torch.float32, use_graph=True, lr=0.0, accumulate=3)
    for c in range(2):
        for b in batches:
            ts(*b)
        assert float((ts.flat_g - gsum).norm() / gsum.norm()) < 2e-2, c
    assert ts.steps_done == 2 + 2                                    # two eager warm-up cycles precede the capture
    assert set(ts.graphs) == {(True, False), (False, False), (False, True)} and torch.equal(ts.flat_p, w0)
    ops.set_step_context(None)
    # ... and with a learning rate the replayed cycles keep stepping
    ts = TrainStep(_model(), dtype=torch.float32, use_graph=True, lr=1e-4, accumulate=2)
    seen = []
    for c in range(3):
        for b in batches[:2]:
            ts(*b)
        seen.append(ts.flat_p.clone())
    ops.set_step_context(None)
    assert set(ts.graphs) == {(True, False), (False, True)} and ts.steps_done == 5
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])
