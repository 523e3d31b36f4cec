"""PSMNet path: 3-D conv / transposed conv, cost volume, fused soft-argmin, and the whole network vs the
reference-captured goldens (tests/golden/psmnet.npz) and the CPU oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import ref_models as R
from oracle.detweights import fill_state_dict, rand_input, randn_input

GDIR = os.path.join(os.path.dirname(__file__), "golden")


def _vol_to_images(x):      # (B,C,D,H,W) -> (B*D, C, H, W) NHWC images on the GPU
    B, C, D, H, W = x.shape
    return x.permute(0, 2, 1, 3, 4).reshape(B * D, C, H, W).contiguous(memory_format=torch.channels_last)


def _images_to_vol(y, B):   # inverse
    BD, C, H, W = y.shape
    return y.reshape(B, BD // B, C, H, W).permute(0, 2, 1, 3, 4)


def _rel(a, b):
    return float((a - b).abs().max()) / max(1e-6, float(b.abs().max()))


def _load_eval_stats(m, gold):
    """The eval fixture carries running statistics collected on its input (oracle/make_golden.py gen_psmnet): those of a
    trained network, so that the softmax over disparities is not saturated."""
    sd = m.state_dict()
    for k in gold.files:
        if k.startswith("psm64.eval.state."):
            sd[k[len("psm64.eval.state."):]].copy_(torch.from_numpy(gold[k]))
    return m


def test_oracle_psmnet_matches_golden():
    gold = np.load(os.path.join(GDIR, "psmnet.npz"))
    m = _load_eval_stats(fill_state_dict(R.PSMNet(64), 41), gold).eval()
    a, b = rand_input(41, "left", (2, 3, 256, 256)), rand_input(41, "right", (2, 3, 256, 256))
    with torch.no_grad():
        p = m(a, b)
    want = gold["psm64.eval.pred0.sample"]
    got = p[:, ::8, ::8].numpy()
    assert got.shape == want.shape and np.abs(got - want).max() < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,stride,D,H,W", [(16, 32, 1, 6, 10, 18), (32, 64, 2, 8, 12, 20), (64, 32, 1, 5, 9, 17), (32, 1, 1, 6, 8, 16)])
def test_hip_conv3d_matches_torch(cin, cout, stride, D, H, W):
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    torch.manual_seed(0)
    ref = nn.Conv3d(cin, cout, 3, stride=stride, padding=1, bias=False)
    x = randn_input(51, "x", (2, cin, D, H, W)).requires_grad_(True)
    y = ref(x)
    g = randn_input(52, "g", tuple(y.shape))
    y.backward(g)
    w = ref.weight.detach().clone().cuda().requires_grad_(True)
    xd = _vol_to_images(x.detach().cuda()).requires_grad_(True)
    yd, Do = ops.conv3d(xd, D, w, stride, 1)
    assert Do == y.shape[2]
    yd.backward(_vol_to_images(g.cuda()))
    assert _rel(_images_to_vol(yd, 2).cpu(), y.detach()) < 2e-4
    assert _rel(_images_to_vol(xd.grad, 2).cpu(), x.grad) < 2e-4
    assert _rel(w.grad.cpu(), ref.weight.grad) < 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("B,co,D,H,W", [(2, 32, 11, 19, 45), (1, 32, 48, 64, 128), (2, 16, 3, 8, 32)])
def test_hip_conv3d_bf16_single_map_head(B, co, D, H, W):
    """The 3x3x3 convolution to ONE map that ends classif1-3 (models/stackhourglass.py:90-102) in bf16: its data gradient is
    the fan-out kernel over volumes (conv_thin.h: 27 taps as the MFMA reduction axis, several depth slices per workgroup);
    forward, data gradient and weight gradient against f32 ATen on the same bf16-rounded operands."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    g = torch.Generator().manual_seed(B * 100 + D)
    x = torch.randn(B, co, D, H, W, generator=g).bfloat16()
    w = (torch.randn(1, co, 3, 3, 3, generator=g) * 0.1)
    gy = torch.randn(B, 1, D, H, W, generator=g).bfloat16()
    xr = x.float().requires_grad_(True)
    wr = w.bfloat16().float().requires_grad_(True)
    yr = F.conv3d(xr, wr, None, padding=1)
    yr.backward(gy.float())
    wd = w.cuda().requires_grad_(True)
    xd = _vol_to_images(x.cuda()).requires_grad_(True)
    yd, Do = ops.conv3d(xd, D, wd, 1, 1)
    assert Do == D and yd.dtype == torch.bfloat16
    yd.backward(_vol_to_images(gy.cuda()))
    torch.cuda.synchronize()
    tol = 2.0 ** -7
    assert _rel(_images_to_vol(yd, B).float().cpu(), yr.detach()) < tol
    assert _rel(_images_to_vol(xd.grad, B).float().cpu(), xr.grad) < tol
    assert _rel(wd.grad.float().cpu(), wr.grad) < tol


@pytest.mark.gpu
@pytest.mark.parametrize("B,ci,co,D,H,W", [(2, 32, 32, 7, 70, 100), (1, 24, 32, 48, 64, 128), (4, 32, 24, 5, 33, 130)])
def test_hip_conv3d_bf16_persistent_volume_kernel(B, ci, co, D, H, W):
    """3x3x3, <= 32 channels on both sides, bf16 on the persistent kernel (conv_band.h with depth taps as chunks: PSMNet's
    32 -> 32 stack): forward and data gradient against f32 ATen on the same bf16-rounded operands; convbn_3d + ReLU + skip in
    training mode (statistics from the epilogue, the data gradient accumulated onto a parked skip gradient) against the
    halo-tile kernel (SDHIP_CONV_NO_BAND3): outputs, gradients and running statistics."""
    import os
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops, _lib
    g = torch.Generator().manual_seed(B * 100 + D)
    x = torch.randn(B, ci, D, H, W, generator=g).bfloat16()
    w = (torch.randn(co, ci, 3, 3, 3, generator=g) * 0.05)
    gy = torch.randn(B, co, D, H, W, generator=g).bfloat16()
    xr = x.float().requires_grad_(True)
    yr = F.conv3d(xr, w.bfloat16().float(), None, padding=1)
    yr.backward(gy.float())
    wd = w.cuda().requires_grad_(True)
    xd = _vol_to_images(x.cuda()).requires_grad_(True)
    yd, Do = ops.conv3d(xd, D, wd, 1, 1)
    yd.backward(_vol_to_images(gy.cuda()))
    torch.cuda.synchronize()
    tol = 2.0 ** -7
    assert _rel(_images_to_vol(yd, B).float().cpu(), yr.detach()) < tol
    assert _rel(_images_to_vol(xd.grad, B).float().cpu(), xr.grad) < tol
    if ci != co:
        return
    # convbn_3d + ReLU, then convbn_3d + skip (the shape of dres1, stackhourglass.py:65-68), both kernels
    res = {}
    for sw in ("", "1"):
        if sw:
            os.environ["SDHIP_CONV_NO_BAND3"] = sw
        else:
            os.environ.pop("SDHIP_CONV_NO_BAND3", None)
        _lib.reload_diag()
        torch.manual_seed(5)
        bn1, bn2 = nn.BatchNorm3d(co).cuda().train(), nn.BatchNorm3d(co).cuda().train()
        w1 = w.cuda().requires_grad_(True); w2 = (w * 0.5).cuda().requires_grad_(True)
        x0 = _vol_to_images(x.cuda()).requires_grad_(True)
        slot = ops.GradSlot(exclusive=True)
        h, _ = ops.conv3d_bn_act(x0, D, w1, bn1, act=1, in_slot=slot)
        y, _ = ops.conv3d_bn_act(h, D, w2, bn2, act=0, residual=x0, res_slot=slot)
        y.backward(_vol_to_images(gy.cuda()))
        torch.cuda.synchronize()
        res[sw] = (y.detach().float(), x0.grad.float(), bn1.running_var.clone(), bn2.running_mean.clone(), bn1.weight.grad.clone())
    os.environ.pop("SDHIP_CONV_NO_BAND3", None)
    _lib.reload_diag()
    for a, b in zip(res[""], res["1"]):
        assert _rel(a, b) < 2.0 ** -6, _rel(a, b)


@pytest.mark.gpu
def test_hip_deconv3d_bn_matches_torch():
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    torch.manual_seed(0)
    ct = nn.ConvTranspose3d(32, 16, 3, padding=1, output_padding=1, stride=2, bias=False)
    bn = nn.BatchNorm3d(16)
    x = randn_input(53, "x", (2, 32, 4, 6, 10)).requires_grad_(True)
    y = bn(ct(x))
    g = randn_input(54, "g", tuple(y.shape))
    y.backward(g)
    w = ct.weight.detach().clone().cuda().requires_grad_(True)
    bnd = nn.BatchNorm3d(16).cuda()
    xd = _vol_to_images(x.detach().cuda()).requires_grad_(True)
    yd, Do = ops.deconv3d_s2_bn_act(xd, 4, w, bnd)
    assert Do == 8
    yd.backward(_vol_to_images(g.cuda()))
    assert _rel(_images_to_vol(yd, 2).cpu(), y.detach()) < 5e-4
    assert _rel(_images_to_vol(xd.grad, 2).cpu(), x.grad) < 2e-3
    assert _rel(w.grad.cpu(), ct.weight.grad) < 2e-3
    assert _rel(bnd.running_var.cpu(), bn.running_var) < 1e-4


@pytest.mark.gpu
def test_hip_cost_volume_and_softargmin_match_torch():
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    B, C, H, W, D = 2, 8, 6, 20, 5
    l = randn_input(55, "l", (B, C, H, W)).requires_grad_(True); r = randn_input(55, "r", (B, C, H, W)).requires_grad_(True)
    cost = l.new_zeros(B, 2 * C, D, H, W)
    parts = []
    for i in range(D):
        sl = torch.zeros(B, 2 * C, H, W)
        if i > 0:
            sl = torch.cat((F.pad(l[:, :, :, i:], (i, 0)), F.pad(r[:, :, :, :-i], (i, 0))), 1)
        else:
            sl = torch.cat((l, r), 1)
        parts.append(sl)
    cost = torch.stack(parts, 2)
    g = randn_input(56, "g", tuple(cost.shape))
    cost.backward(g)
    ld = l.detach().cuda().requires_grad_(True); rd = r.detach().cuda().requires_grad_(True)
    vol = ops.cost_volume(ld, rd, D)
    vol.backward(_vol_to_images(g.cuda()))
    assert _rel(_images_to_vol(vol, B).cpu(), cost.detach()) < 1e-6
    assert _rel(ld.grad.cpu(), l.grad) < 1e-5 and _rel(rd.grad.cpu(), r.grad) < 1e-5
    # fused soft-argmin vs upsample -> softmax -> regression
    D4, H4, W4, maxd = 6, 8, 8, 24
    c = randn_input(57, "c", (B, 1, D4, H4, W4)).requires_grad_(True)
    up = F.interpolate(c, [maxd, 32, 32], mode='trilinear').squeeze(1)
    pred = torch.sum(F.softmax(up, 1) * torch.arange(maxd, dtype=torch.float32).view(1, -1, 1, 1), 1)
    gp = randn_input(58, "gp", tuple(pred.shape))
    pred.backward(gp)
    cd = _vol_to_images(c.detach().cuda()).requires_grad_(True)
    pd = ops.soft_argmin(cd, D4, maxd, 32, 32)
    pd.backward(gp.cuda())
    assert float((pd.cpu() - pred.detach()).abs().max()) < 1e-4
    assert _rel(_images_to_vol(cd.grad, B).cpu(), c.grad) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("D4,maxd,H4,W4,H,W", [(6, 24, 8, 8, 24, 20), (5, 17, 8, 8, 24, 20), (16, 64, 9, 7, 36, 28), (3, 12, 5, 6, 20, 24)])
def test_hip_softargmin_scales(D4, maxd, H4, W4, H, W):
    """Soft-argmin at other scale factors than PSMNet's exact x4: depth x4 with non-integer spatial factors (the gather
    backward's footprint search), a depth factor that is not 4 (general kernels), ragged sizes — forward and the gradient
    w.r.t. the low-resolution cost against the unfused ATen sequence (stackhourglass.py:138-155, submodule.py:56-64)."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    B = 2
    c = (randn_input(61, "cs", (B, 1, D4, H4, W4)) * 2.0).requires_grad_(True)
    up = F.interpolate(c, [maxd, H, W], mode='trilinear').squeeze(1)
    pred = torch.sum(F.softmax(up, 1) * torch.arange(maxd, dtype=torch.float32).view(1, -1, 1, 1), 1)
    gp = randn_input(62, "gps", tuple(pred.shape))
    pred.backward(gp)
    cd = _vol_to_images(c.detach().cuda()).requires_grad_(True)
    pd = ops.soft_argmin(cd, D4, maxd, H, W)
    pd.backward(gp.cuda())
    assert float((pd.cpu() - pred.detach()).abs().max()) < 1e-4 * maxd
    assert _rel(_images_to_vol(cd.grad, B).cpu(), c.grad) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_hip_softargmin_maxdisp192(dtype):
    """The 192-level head of BASELINE configs 3-4 (models_psmnet/stackhourglass.py:138-155 with maxdisp = 192): cost
    (B,1,48,H/4,W/4) -> trilinear x4 -> softmax over 192 levels -> expectation, fused, vs the unfused ATen sequence."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    B, D4, H4, W4, maxd = 2, 48, 12, 20, 192
    c = (randn_input(59, "c192", (B, 1, D4, H4, W4)) * 3.0)
    if dtype == torch.bfloat16:
        c = c.bfloat16().float()
    c.requires_grad_(True)
    up = F.interpolate(c, [maxd, 4 * H4, 4 * W4], mode='trilinear').squeeze(1)
    pred = torch.sum(F.softmax(up, 1) * torch.arange(maxd, dtype=torch.float32).view(1, -1, 1, 1), 1)
    gp = randn_input(60, "gp192", tuple(pred.shape))
    pred.backward(gp)
    cd = _vol_to_images(c.detach().cuda().to(dtype)).requires_grad_(True)
    pd = ops.soft_argmin(cd, D4, maxd, 4 * H4, 4 * W4)
    pd.backward(gp.cuda().to(dtype))
    tol = 2e-3 if dtype == torch.float32 else 1.0      # bf16 stores the 0..191 expectation with 8 mantissa bits (ulp 1 at 128+)
    assert float((pd.float().cpu() - pred.detach()).abs().max()) < tol
    assert _rel(_images_to_vol(cd.grad.float(), B).cpu(), c.grad) < (2e-3 if dtype == torch.float32 else 3e-2)


@pytest.mark.gpu
def test_hip_psmnet192_matches_cpu_oracle():
    """PSMNet(192) (util/utilLoadNetwork.py:52-54) train mode, B = 2, 256x256: three predictions and the mean-L1 loss
    against the CPU oracle (pinned by the PSMNet(64) fixtures), f32, 1e-3 of the disparity range."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.psmnet import PSMNet
    torch.set_num_threads(16)
    a, b = rand_input(45, "left", (2, 3, 256, 256)), rand_input(45, "right", (2, 3, 256, 256))
    disp = rand_input(45, "disp", (2, 256, 256), 0.0, 150.0)
    ref = fill_state_dict(R.PSMNet(192), 45).train()
    with torch.no_grad():
        want = ref(a, b)
    m = fill_state_dict(PSMNet(192), 45).cuda().train()
    outs = m(a.cuda(), b.cuda())
    loss = sum(F.l1_loss(o, disp.cuda()) for o in outs) / 3
    loss.backward()
    for o, w in zip(outs, want):
        # THE GATE is the relative bound (max-abs error over max-abs value, 1e-3: north star); the absolute line only says the
        # same thing in pixels for a reader (1e-3 of the 192-level range = 0.19 px) and can never be the tighter one
        assert _rel(o.detach().cpu(), w) < 1e-3
        assert float((o.detach().cpu() - w).abs().max()) <= 1e-3 * 192
    wl = float(sum(F.l1_loss(w, disp) for w in want) / 3)
    assert abs(float(loss) - wl) <= 1e-3 * wl
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_hip_psmnet_matches_golden(mode):
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.psmnet import PSMNet
    gold = np.load(os.path.join(GDIR, "psmnet.npz"))
    m = fill_state_dict(PSMNet(64), 41)
    if mode == "eval":
        _load_eval_stats(m, gold)
    m = m.cuda()
    m.train() if mode == "train" else m.eval()
    a, b = rand_input(41, "left", (2, 3, 256, 256)).cuda(), rand_input(41, "right", (2, 3, 256, 256)).cuda()
    disp = rand_input(41, "disp", (2, 256, 256), 0.0, 40.0).cuda()
    outs = m(a, b)
    outs = outs if isinstance(outs, tuple) else (outs,)
    loss = sum(F.l1_loss(o, disp) for o in outs) / len(outs)
    loss.backward()
    p = "psm64.%s" % mode
    for i, o in enumerate(outs):
        want = gold["%s.pred%d.sample" % (p, i)]
        got = o.detach().cpu()[:, ::8, ::8].numpy()
        assert got.shape == want.shape
        err = np.abs(got - want)
        assert err.max() <= 1e-3 * max(1.0, np.abs(want).max()), (mode, i, err.max())
    tol = 1e-3
    assert abs(loss.item() - float(gold[p + ".loss"])) <= tol * max(1.0, float(gold[p + ".loss"]))
    acc = {}
    for k, q in m.named_parameters():
        if q.grad is not None:
            top = k.split(".")[0]
            acc[top] = acc.get(top, 0.0) + float(q.grad.double().pow(2).sum())
    for top, v in acc.items():
        key = "%s.gnorm.%s" % (p, top)
        if key in gold.files and mode == "train":
            w = float(gold[key])
            assert abs(np.sqrt(v) - w) <= 3e-2 * max(w, 1e-6), (key, np.sqrt(v), w)


@pytest.mark.gpu
def test_hip_psmnet192_config4_properties():
    """BASELINE config 4 at its stated workload: PSMNet(192) (util/utilLoadNetwork.py:52-54), 960 x 512 (W x H), batch 4 per
    GPU, bf16, the captured training step (cost volume (4,64,48,128,240): models_psmnet/stackhourglass.py:110-119).  Shapes,
    finite predictions in the disparity range, finite gradients for every parameter, a loss that falls over optimizer steps
    on one batch and replays that are reproducible at lr 0."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.psmnet import PSMNet
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch
    B, H, W = 4, 512, 960
    torch.manual_seed(0)
    m = PSMNet(192).cuda().train()
    left, right, seg, disp = synthetic_batch(B, H, W, seed=9)
    disp = disp * 10.0                                   # spread the targets over the disparity range
    with torch.no_grad():
        outs = m(left.bfloat16(), right.bfloat16())
    assert len(outs) == 3
    for o in outs:
        assert tuple(o.shape) == (B, H, W) and torch.isfinite(o.float()).all()
        assert float(o.float().min()) >= 0.0 and float(o.float().max()) <= 191.0 + 1.0       # an expectation over 0..191 (bf16 rounding)
    loss_fn = lambda outs, seg, disp: ops.mean_l1_loss(outs, disp[:, 0])
    ts = TrainStep(m, dtype=torch.bfloat16, use_graph=True, lr=1e-3, loss_fn=loss_fn)
    losses = [float(ts(left, right, seg, disp)) for _ in range(6)]
    assert ts.graph is not None and all(np.isfinite(losses)), losses
    assert torch.isfinite(ts.flat_g).all() and float(ts.flat_g.norm()) > 0
    assert min(losses[3:]) < losses[0], losses
    ops.set_step_context(None)
    del ts, m
    torch.cuda.empty_cache()
