"""conv2dSame / ConvTranspose2dSame / convbn / deconvbn / Conv2DownUp: HIP kernels vs the reference-captured
golden vectors (tests/golden/ops.npz) and vs the CPU oracle on larger, ragged and bf16 cases."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_models as R
from oracle.detweights import fill_state_dict, randn_input

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ops.npz")

CONV_CASES = [("c1", 8, 16, 1, 1, 1, 9, 11), ("c3", 16, 8, 3, 1, 1, 12, 17), ("c5", 8, 8, 5, 1, 1, 13, 16),
              ("c7s2", 3, 8, 7, 2, 1, 17, 20), ("c5d2", 3, 1, 5, 1, 2, 16, 19), ("c3s2", 8, 8, 3, 2, 1, 15, 16)]
DECONV_CASES = [("d3", 8, 16, 3, 1, 12, 17), ("d5", 8, 4, 5, 1, 13, 16)]


def _close(got, want, tol, what=""):
    got = got.detach().float().cpu()
    want = torch.as_tensor(want).float()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = max(1.0, float(want.abs().max()))
    err = float((got - want).abs().max())
    assert err <= tol * scale, "%s: max err %.3e > %.1e * %.3g" % (what, err, tol, scale)


# ------------------------------------------------------------------ CPU: the oracle reproduces the golden vectors
@pytest.mark.parametrize("case", CONV_CASES)
def test_oracle_conv_matches_golden(case):
    name, ci, co, k, s, d, H, W = case
    gold = np.load(GOLD)
    m = fill_state_dict(R.conv2dSame(ci, co, k, s, 'same', d, bias=True), 11)
    x = randn_input(11, name, (2, ci, H, W)).requires_grad_(True)
    y = m(x)
    y.backward(randn_input(12, name, tuple(y.shape)))
    np.testing.assert_allclose(y.detach().numpy(), gold["conv.%s.y" % name], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(x.grad.numpy(), gold["conv.%s.gx" % name], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(m.c2d.weight.grad.numpy(), gold["conv.%s.gw" % name], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name,ctor,shape", [
    ("convbn", lambda: R.convbn(8, 16, 3, 1, 'same', 1), (4, 8, 10, 12)),
    ("deconvbn", lambda: R.deconvbn(8, 8, 5, 1, 'same', 1), (4, 8, 10, 12)),
    ("cdu_last", lambda: R.Conv2DownUp(8, 16, 3, True), (2, 8, 12, 16)),
    ("cdu_nolast", lambda: R.Conv2DownUp(16, 8, 5, False), (2, 16, 12, 16))])
def test_oracle_blocks_match_golden(name, ctor, shape):
    gold = np.load(GOLD)
    m = fill_state_dict(ctor(), 15).train()
    x = randn_input(15, name, shape).requires_grad_(True)
    y = m(x)
    y.backward(randn_input(16, name, tuple(y.shape)))
    np.testing.assert_allclose(y.detach().numpy(), gold["%s.y" % name], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(x.grad.numpy(), gold["%s.gx" % name], rtol=1e-3, atol=1e-4)
    for k, v in m.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            np.testing.assert_allclose(v.numpy(), gold["%s.state.%s" % (name, k)], rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------ GPU: HIP path vs golden / oracle
@pytest.mark.gpu
@pytest.mark.parametrize("case", CONV_CASES)
def test_hip_conv_matches_golden(case):
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N
    name, ci, co, k, s, d, H, W = case
    gold = np.load(GOLD)
    m = fill_state_dict(N.conv2dSame(ci, co, k, s, 'same', d, bias=True), 11).cuda()
    x = randn_input(11, name, (2, ci, H, W)).cuda().requires_grad_(s == 1)
    y = m(x)
    _close(y, gold["conv.%s.y" % name], 1e-4, name + ".y")
    y.backward(randn_input(12, name, tuple(y.shape)).cuda())
    if s == 1:
        _close(x.grad, gold["conv.%s.gx" % name], 1e-4, name + ".gx")
    _close(m.c2d.weight.grad, gold["conv.%s.gw" % name], 1e-4, name + ".gw")
    _close(m.c2d.bias.grad, gold["conv.%s.gb" % name], 1e-4, name + ".gb")


@pytest.mark.gpu
@pytest.mark.parametrize("case", DECONV_CASES)
def test_hip_deconv_matches_golden(case):
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N
    name, ci, co, k, s, H, W = case
    gold = np.load(GOLD)
    m = fill_state_dict(N.ConvTranspose2dSame(ci, co, k, s, 'same', 1, bias=True), 13).cuda()
    x = randn_input(13, name, (2, ci, H, W)).cuda().requires_grad_(True)
    y = m(x)
    _close(y, gold["deconv.%s.y" % name], 1e-4, name + ".y")
    y.backward(randn_input(14, name, tuple(y.shape)).cuda())
    _close(x.grad, gold["deconv.%s.gx" % name], 1e-4, name + ".gx")
    _close(m.ct2d.weight.grad, gold["deconv.%s.gw" % name], 1e-4, name + ".gw")
    _close(m.ct2d.bias.grad, gold["deconv.%s.gb" % name], 1e-4, name + ".gb")


@pytest.mark.gpu
@pytest.mark.parametrize("k,s,ci,co", [(3, 2, 8, 8), (5, 2, 16, 8), (4, 2, 8, 16)])
def test_hip_strided_deconv_matches_oracle(k, s, ci, co):
    """models/torch_model.py:284-349 with stride 2 (the dsnet decoder): full transposed conv + centre crop."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N
    from oracle import ref_models as R
    ref = fill_state_dict(R.ConvTranspose2dSame(ci, co, k, s, 'same', 1, bias=True), 17)
    m = fill_state_dict(N.ConvTranspose2dSame(ci, co, k, s, 'same', 1, bias=True), 17).cuda()
    x0 = randn_input(17, "sdeconv", (2, ci, 6, 10))
    xr = x0.clone().requires_grad_(True)
    yr = ref(xr)
    gy = randn_input(18, "sdeconv", tuple(yr.shape))
    yr.backward(gy)
    x = x0.cuda().requires_grad_(True)
    y = m(x)
    assert tuple(y.shape) == tuple(yr.shape)
    _close(y, yr.detach().numpy(), 1e-4, "sdeconv.y")
    y.backward(gy.cuda())
    _close(x.grad, xr.grad.numpy(), 1e-4, "sdeconv.gx")
    _close(m.ct2d.weight.grad, ref.ct2d.weight.grad.numpy(), 1e-4, "sdeconv.gw")
    _close(m.ct2d.bias.grad, ref.ct2d.bias.grad.numpy(), 1e-4, "sdeconv.gb")


@pytest.mark.gpu
@pytest.mark.parametrize("name,ctor,shape", [
    ("convbn", lambda N: N.convbn(8, 16, 3, 1, 'same', 1), (4, 8, 10, 12)),
    ("deconvbn", lambda N: N.deconvbn(8, 8, 5, 1, 'same', 1), (4, 8, 10, 12)),
    ("cdu_last", lambda N: N.Conv2DownUp(8, 16, 3, True), (2, 8, 12, 16)),
    ("cdu_nolast", lambda N: N.Conv2DownUp(16, 8, 5, False), (2, 16, 12, 16))])
def test_hip_blocks_match_golden(name, ctor, shape):
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N
    gold = np.load(GOLD)
    m = fill_state_dict(ctor(N), 15).cuda().train()
    x = randn_input(15, name, shape).cuda().requires_grad_(True)
    y = m(x)
    _close(y, gold["%s.y" % name], 2e-4, name + ".y")
    y.backward(randn_input(16, name, tuple(y.shape)).cuda())
    _close(x.grad, gold["%s.gx" % name], 1e-3, name + ".gx")
    for k, v in m.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            _close(v, gold["%s.state.%s" % (name, k)], 1e-4, name + "." + k)
    for k, p in m.named_parameters():
        gk = "%s.grad.%s" % (name, k)
        if gk in gold.files:
            _close(p.grad, gold[gk], 1e-3, gk)


BIG = [  # name, cin, cout, k, dil, kind, H, W, B
    ("k5_64", 64, 64, 5, 1, 'conv', 40, 70, 2),       # the hot shape (Conv2DownUp5), ragged tiles
    ("k3_128_64", 128, 64, 3, 1, 'conv', 33, 47, 2),  # two channel chunks
    ("k1_65_64", 65, 64, 1, 1, 'conv', 24, 40, 2),    # ragged Cin (scalar staging path)
    ("k1_2048_64", 2048, 64, 1, 1, 'conv', 8, 16, 2),
    ("k3_32_2", 32, 2, 3, 1, 'deconv', 32, 48, 2),    # segmentation head, Cout = labels
    ("k5_64_1", 64, 1, 5, 1, 'deconv', 32, 48, 2),    # disparity head, Cout = 1
    ("k3_d6", 32, 48, 3, 6, 'conv', 20, 36, 2),       # ASPP-style dilation (plain padding = dilation)
    ("k3_32_32", 32, 32, 3, 1, 'deconv', 64, 128, 4),
]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 4e-2)])
@pytest.mark.parametrize("case", BIG)
def test_hip_conv_matches_oracle(case, dtype, tol):
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N
    name, ci, co, k, dil, kind, H, W, B = case
    if kind == 'conv':
        ref = R.conv2dSame(ci, co, k, 1, 'same', dil, bias=True)
        mine = N.conv2dSame(ci, co, k, 1, 'same', dil, bias=True)
    else:
        ref = R.ConvTranspose2dSame(ci, co, k, 1, 'same', 1, bias=True)
        mine = N.ConvTranspose2dSame(ci, co, k, 1, 'same', 1, bias=True)
    fill_state_dict(ref, 41)
    mine.load_state_dict(ref.state_dict())
    x = randn_input(41, name, (B, ci, H, W))
    g = randn_input(42, name, (B, co, H, W))
    if dtype == torch.bfloat16:  # same rounded operands on both sides; weights rounded like the kernel does
        x = x.bfloat16().float(); g = g.bfloat16().float()
        with torch.no_grad():
            for p in ref.parameters():
                if p.dim() > 1:
                    p.copy_(p.bfloat16().float())
        mine.load_state_dict(ref.state_dict())
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    yr.backward(g)
    mine = mine.cuda()
    xd = x.cuda().to(dtype).requires_grad_(True)
    y = mine(xd)
    assert y.dtype == dtype and y.shape == yr.shape
    y.backward(g.cuda().to(dtype))
    _close(y, yr.detach(), tol, name + ".y")
    _close(xd.grad, xr.grad, tol, name + ".gx")
    rp = dict(ref.named_parameters())
    for kname, p in mine.named_parameters():
        _close(p.grad, rp[kname].grad, tol, name + "." + kname)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,Cin,ldx,Cout,pro,groups", [
    (4, 16, 32, 1024, 1024, 128, True, 2), (4, 16, 32, 992, 1024, 128, True, 1), (2, 32, 64, 512, 512, 256, True, 1),
    (2, 32, 48, 192, 256, 128, True, 2), (2, 24, 40, 128, 128, 128, False, 1), (2, 20, 36, 96, 256, 128, True, 1),
    (2, 16, 16, 160, 160, 64, False, 1), (1, 7, 9, 320, 320, 48, True, 1), (2, 16, 32, 65, 72, 64, False, 1)])
def test_hip_1x1_wgrad_many_channels(B, H, W, Cin, ldx, Cout, pro, groups):
    """The chunk-packed 1x1 weight gradient (conv_wgrad_fast.h, WgfArgs::qb): DenseNet bottleneck / transition shapes —
    input = a channel prefix of a wider slab (ldx > Cin), BatchNorm+ReLU prologue per statistics group, ragged tiles,
    partial last chunk — against an f32 contraction of the same (bf16-rounded) operands."""
    import os
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops, _lib
    os.environ["SDHIP_WGRAD_FORCE_PACK"] = "1"      # the launch heuristic would keep these small maps on the unpacked kernel
    _lib.reload_diag()
    torch.manual_seed(Cin + Cout)
    dev = torch.device("cuda:0")
    slab = torch.randn(B, H, W, ldx, device=dev).to(torch.bfloat16)
    x = slab.permute(0, 3, 1, 2)[:, :Cin]
    g = (torch.randn(B, H, W, Cout, device=dev) * 0.1).to(torch.bfloat16).permute(0, 3, 1, 2)
    w = torch.zeros(Cout, Cin, 1, 1, device=dev)
    sc = (torch.rand(groups, Cin, device=dev) + 0.5) if pro else None
    sh = (torch.rand(groups, Cin, device=dev) - 0.5) if pro else None
    spec = ops.ConvSpec('conv', 1, 1, 1, 1, 0, 0, H, W)
    ops.set_step_context(None)
    gw, _ = ops._wgrad_impl(x, ldx, g, Cout, w, None, spec, sc, sh, pro, groups)
    xe = x.float()
    if pro:
        per = B // groups
        s4 = sc.repeat_interleave(per, 0)[:, :, None, None]
        h4 = sh.repeat_interleave(per, 0)[:, :, None, None]
        xe = torch.relu(torch.addcmul(h4, xe, s4)).to(torch.bfloat16).float()     # the kernel rounds the prologue result to bf16
    want = torch.einsum("bmhw,bchw->mc", g.float(), xe)
    os.environ.pop("SDHIP_WGRAD_FORCE_PACK", None)
    _lib.reload_diag()
    err = float((gw.reshape(Cout, Cin) - want).norm() / want.norm())
    assert err < 2e-3, err


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,Cin,bias", [(2, 128, 256, 8, True), (2, 130, 250, 8, False), (1, 256, 512, 3, True)])
def test_hip_thin_wgrad_tiled(B, H, W, Cin, bias):
    """5x5 dilation-2 'same' convolution of an <= 8-channel image to ONE channel (conv2d_ba* of models/dsnet_t2.py): the
    LDS-tiled weight-gradient kernel (conv_thin.h) on full-size and ragged maps against autograd on the same bf16-rounded
    operands."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    torch.manual_seed(B * H + Cin)
    dev = torch.device("cuda:0")
    x8 = torch.zeros(B, H, W, 8, device=dev, dtype=torch.bfloat16)
    x8[..., :Cin] = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
    x = x8.permute(0, 3, 1, 2)[:, :Cin]
    g1 = (torch.randn(B, H, W, 1, device=dev) * 0.1).to(torch.bfloat16).permute(0, 3, 1, 2)
    w = torch.zeros(1, Cin, 5, 5, device=dev)
    b = torch.zeros(1, device=dev) if bias else None
    spec = ops.ConvSpec('conv', 5, 5, 1, 2, 4, 4, H, W)
    ops.set_step_context(None)
    gw, gb = ops._wgrad_impl(x, 8, g1, 1, w, b, spec, None, None, False, 1)
    wr = torch.zeros(1, Cin, 5, 5, device=dev, requires_grad=True)
    y = torch.nn.functional.conv2d(x.float(), wr, None, 1, 4, 2)
    (y * g1.float()).sum().backward()
    err = float((gw - wr.grad).norm() / wr.grad.norm())
    assert err < 1e-4, err
    if bias:
        assert abs(float(gb) - float(g1.float().sum())) <= 1e-3 * max(1.0, abs(float(g1.float().sum())))


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,Cin,Cout,groups", [(4, 64, 128, 32, 128, 2), (2, 33, 47, 32, 128, 1), (4, 16, 32, 32, 128, 2),
                                                   (2, 20, 36, 64, 64, 1)])
def test_hip_dgrad_with_bn_backward_sums(B, H, W, Cin, Cout, groups):
    """sdhip_conv2d_fwd_bnbwd (the 3x3 data gradient of a DenseNet layer with norm2's backward reductions in its epilogue):
    output bit-identical to the plain launch, sums equal to sdhip_affine_act_bwd's over that output."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    from pmt_learning_for_semantic_segmentation_and_disparity_amd._lib import call, ptr, stream_ptr, dtype_code
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(B * 100 + H)
    x = torch.randn(B, H, W, Cin, generator=g).to(dev).bfloat16().permute(0, 3, 1, 2)
    u = torch.randn(B, H, W, Cout, generator=g).to(dev).bfloat16().permute(0, 3, 1, 2)
    w = (torch.randn(Cin, Cout, 3, 3, generator=g) * 0.1).to(dev)             # forward conv Cout -> Cin; this is its data gradient
    sc = (torch.rand(groups, Cout, generator=g) + 0.5).to(dev)
    sh = (torch.randn(groups, Cout, generator=g) * 0.3).to(dev)
    wd = ops.packed_weight(w, 'conv', 'dgrad', torch.bfloat16)
    dt = dtype_code(x)
    y0 = ops.empty_nhwc(B, Cout, H, W, torch.bfloat16, dev)
    ops._conv_launch(x, Cin, wd, y0, Cout, None, None, None, None, B, H, W, Cin, H, W, Cout, 3, 3, 1, 1, 1, 1, False, 1, 0, False)
    both = torch.zeros(2, ops.NREP, groups, Cout, dtype=torch.float32, device=dev)
    call("sdhip_affine_act_bwd", ptr(y0), Cout, ptr(u), Cout, None, 0, ptr(sc), ptr(sh), ptr(both[0]), ptr(both[1]), ops.NREP,
         B * H * W, Cout, groups, 1, 0, 0, dt, stream_ptr())
    y1 = ops.empty_nhwc(B, Cout, H, W, torch.bfloat16, dev)
    sums = torch.zeros(ops.NREP, groups, 2, Cout, dtype=torch.float64, device=dev)
    call("sdhip_conv2d_fwd_bnbwd", ptr(x), ptr(wd), ptr(y1), ptr(sums), Cout, ops.NREP, ptr(u), Cout, ptr(sc), ptr(sh), None, 0,
         B, H, W, Cin, Cin, H, W, Cout, Cout, 3, 3, 1, 1, 1, groups, 0, dt, stream_ptr())
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    ref = both.double().sum(1)                     # [2][groups][C]
    got = sums.sum(0).permute(1, 0, 2)             # [2][groups][C]
    scale = ref.abs().max().item()
    assert (ref - got).abs().max().item() <= 2e-4 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,K,C,Ct,groups", [(4, 16, 32, 128, 256, 512, 2), (2, 8, 16, 128, 992, 1024, 1), (4, 20, 36, 128, 96, 160, 2)])
def test_hip_1x1_dgrad_apply_mode(B, H, W, K, C, Ct, groups):
    """Mode 1 of sdhip_conv2d_fwd_bnbwd (DenseNet: the 1x1 data gradient that also performs the first phase of norm1's
    backward): slab gradient and reductions equal the two separate launches (conv, then sdhip_affine_act_bwd with
    accumulate) up to the rounding of the intermediate the fused launch never stores."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    from pmt_learning_for_semantic_segmentation_and_disparity_amd._lib import call, ptr, stream_ptr, dtype_code
    dev = "cuda"
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(B, H, W, K, generator=g).to(dev).bfloat16().permute(0, 3, 1, 2)              # gradient of the 1x1 conv's output
    slab = torch.randn(B, H, W, Ct, generator=g).to(dev).bfloat16().permute(0, 3, 1, 2)          # features (u = first C channels)
    gs0 = torch.randn(B, H, W, Ct, generator=g).to(dev).bfloat16().permute(0, 3, 1, 2)           # slab gradient so far
    w = (torch.randn(K, C, 1, 1, generator=g) * 0.1).to(dev)                                      # forward conv C -> K
    sc = (torch.rand(groups, C, generator=g) + 0.5).to(dev)
    sh = (torch.randn(groups, C, generator=g) * 0.3).to(dev)
    wd = ops.packed_weight(w, 'conv', 'dgrad', torch.bfloat16)
    dt = dtype_code(x)
    npix = B * H * W
    # separate launches
    gp = ops.empty_nhwc(B, C, H, W, torch.bfloat16, dev)
    ops._conv_launch(x, K, wd, gp, C, None, None, None, None, B, H, W, K, H, W, C, 1, 1, 1, 1, 0, 0, False, 1, 0, False)
    gs_ref = gs0.clone(memory_format=torch.preserve_format)
    both0 = torch.zeros(2, ops.NREP, groups, C, dtype=torch.float32, device=dev)
    call("sdhip_affine_act_bwd", ptr(gp), C, ptr(slab), Ct, ptr(gs_ref), Ct, ptr(sc), ptr(sh), ptr(both0[0]), ptr(both0[1]), ops.NREP,
         npix, C, groups, 1, 1, 0, dt, stream_ptr())
    # fused
    gs = gs0.clone(memory_format=torch.preserve_format)
    both = torch.zeros(2, ops.NREP, groups, C, dtype=torch.float32, device=dev)
    call("sdhip_conv2d_fwd_bnbwd", ptr(x), ptr(wd), ptr(gs), ptr(both), C, ops.NREP, ptr(slab), Ct, ptr(sc), ptr(sh), ptr(gs), Ct,
         B, H, W, K, K, H, W, C, Ct, 1, 1, 1, 0, 0, groups, 1, dt, stream_ptr())
    torch.cuda.synchronize()
    a, b = gs.float()[:, :C], gs_ref.float()[:, :C]
    assert (a - b).abs().max().item() <= 3e-2 * b.abs().max().item()
    assert torch.equal(gs.float()[:, C:], gs0.float()[:, C:])                 # channels beyond C untouched
    ra, rb = both.double().sum(1), both0.double().sum(1)
    assert (ra - rb).abs().max().item() <= 2e-2 * rb.abs().max().item()
