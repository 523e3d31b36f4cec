"""The persistent whole-CU 5x5 kernel (csrc/conv_band.h) against an f32 contraction of the same bf16-rounded operands
(torch on the CPU), and against the halo-tile kernel it replaces (SDHIP_CONV_NO_BAND=1) bit for bit where both exist.
Shapes are chosen so that the dispatch of conv2d_fwd_impl takes the band path (>= 192 tiles of 16 x 32 pixels)."""
import os
import zlib

import pytest
import torch
import torch.nn.functional as F

CASES = [  # name, B, Cin, Cout, H, W, kind, act
    ("c64_64_ragged", 2, 64, 64, 100, 450, 'conv', 0),      # the hot shape, ragged tiles in both directions
    ("c32_32", 4, 32, 32, 96, 256, 'conv', 1),              # one 32-channel half, 32 output channels (BN = 32)
    ("c16_64", 2, 16, 64, 128, 416, 'conv', 0),             # half-filled channel half
    ("c64_32", 2, 64, 32, 128, 400, 'conv', 2),             # two halves, BN = 32, sigmoid epilogue
    ("c64_48", 2, 64, 48, 128, 400, 'conv', 0),             # Cout not a multiple of the block: general store path
    ("d64_64", 2, 64, 64, 112, 448, 'deconv', 0),           # ConvTranspose2dSame: flipped taps, asymmetric crop
    # 3x3, <= 32 input channels: one stage per chunk, weights resident in LDS
    ("k3_32_32", 4, 32, 32, 96, 256, 'conv', 1, 3),
    ("k3_16_64_ragged", 2, 16, 64, 100, 450, 'conv', 0, 3),
    ("k3_d32_32", 2, 32, 32, 112, 448, 'deconv', 0, 3),
    ("k3_32_48", 2, 32, 48, 128, 400, 'conv', 2, 3),
]


def _ref_conv(x, w, b, kind, act):
    pad = w.shape[-1] // 2
    if kind == 'conv':
        y = F.conv2d(x, w, b, padding=pad)
    else:
        y = F.conv_transpose2d(x, w, b, padding=pad)
    if act == 1:
        y = torch.relu(y)
    elif act == 2:
        y = torch.sigmoid(y)
    return y


def _rel(a, b):
    return ((a.float().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_band_conv_matches_f32_reference(case):
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    name, B, ci, co, H, W, kind, act = case[:8]
    k = case[8] if len(case) > 8 else 5
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 10007)
    x = torch.randn(B, ci, H, W, generator=g).bfloat16().float()
    wshape = (co, ci, k, k) if kind == 'conv' else (ci, co, k, k)
    w = (torch.randn(wshape, generator=g) * 0.05).bfloat16().float()
    b = torch.randn(co, generator=g)
    gy = torch.randn(B, co, H, W, generator=g).bfloat16().float()
    xd = x.cuda().bfloat16().requires_grad_(True)
    wd = w.cuda().requires_grad_(True)
    bd = b.cuda().requires_grad_(True)
    y = ops.conv2d(xd, wd, bd, kind=kind, padding='same' if kind == 'conv' else 'ctsame', act=act)
    y.backward(gy.cuda().bfloat16())
    torch.cuda.synchronize()
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = _ref_conv(xr, wr, br, kind, act)
    if act == 1:
        # ReLU: the derivative is taken where the STORED output is positive; an output within f32 rounding of zero may fall
        # on the other side in the reference's summation order, and one flipped pixel moves gx by |gy * w| ~ 1e-2 of its
        # range — the reference therefore back-propagates through the device's own mask
        ypre = _ref_conv(xr, wr, br, kind, 0)
        ypre.backward(gy * (y.detach().float().cpu() > 0))
    else:
        yr.backward(gy)
    assert _rel(y, yr.detach()) < 1e-2, name           # bf16 rounding of the stored output only
    assert _rel(xd.grad, xr.grad) < 1e-2, name
    assert _rel(wd.grad, wr.grad) < 1e-2, name
    assert _rel(bd.grad, br.grad) < 1e-2, name


@pytest.mark.gpu
@pytest.mark.parametrize("B,ci,co,H,W,groups", [(4, 64, 64, 96, 256, 2), (6, 32, 64, 80, 224, 3), (2, 64, 32, 100, 450, 1)])
def test_band_conv_statistics_match_old_kernel(B, ci, co, H, W, groups):
    """conv + BatchNorm(training) + ReLU with the statistics taken in the conv epilogue: per-group sums of the band kernel
    (accumulated over the tiles of a workgroup, flushed at every group change) give the same normalised output and running
    statistics as an f32 reference, and the convolution itself is bit-identical to the halo-tile kernel's."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops, _lib
    g = torch.Generator().manual_seed(B * 1000 + ci + co)
    x = torch.randn(B, ci, H, W, generator=g).bfloat16().float()
    w = (torch.randn(co, ci, 5, 5, generator=g) * 0.05).bfloat16().float()
    bn = torch.nn.BatchNorm2d(co)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(co, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(co, generator=g) * 0.1)
    # reference: per statistics group (contiguous batch slices) training-mode BatchNorm of the bf16-rounded conv output
    yc = F.conv2d(x, w, None, padding=2).bfloat16().float()
    outs = []
    for yg in yc.chunk(groups, 0):
        m = yg.mean((0, 2, 3), keepdim=True)
        v = yg.var((0, 2, 3), unbiased=False, keepdim=True)
        outs.append(torch.relu((yg - m) / torch.sqrt(v + bn.eps) * bn.weight.view(1, -1, 1, 1) + bn.bias.view(1, -1, 1, 1)))
    yr = torch.cat(outs, 0)

    def run():
        bnd = torch.nn.BatchNorm2d(co).cuda()
        bnd.load_state_dict(bn.state_dict())
        bnd.train()
        xd = x.cuda().bfloat16()
        y = ops.conv_bn_act(xd, w.cuda(), bnd, padding='same', act=1, groups=groups)
        yplain = ops.conv2d(xd, w.cuda(), None, padding='same')
        torch.cuda.synchronize()
        return y.float().cpu(), yplain.float().cpu(), bnd.running_mean.cpu().clone(), bnd.running_var.cpu().clone()

    y, yplain, rm, rv = run()
    assert ((y - yr).abs().max() / yr.abs().max()).item() < 2e-2
    old = os.environ.get("SDHIP_CONV_NO_BAND")
    os.environ["SDHIP_CONV_NO_BAND"] = "1"
    _lib.reload_diag()
    try:
        y0, yplain0, rm0, rv0 = run()
    finally:
        if old is None:
            os.environ.pop("SDHIP_CONV_NO_BAND", None)
        else:
            os.environ["SDHIP_CONV_NO_BAND"] = old
        _lib.reload_diag()
    # same products, different summation order over (tap, channel half): equal up to f32 rounding before the bf16 store
    assert (yplain - yplain0).abs().max().item() <= 2e-2 * yplain0.abs().max().item()
    assert ((yplain != yplain0).float().mean().item()) < 0.2
    assert torch.allclose(rm, rm0, rtol=1e-3, atol=1e-4) and torch.allclose(rv, rv0, rtol=1e-3, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("B,ci,co,H,W", [(2, 64, 64, 100, 450), (4, 32, 64, 96, 256)])
def test_band_conv_sums_a_second_tensor(B, ci, co, H, W):
    """y = conv(x) + addend (sdhip_conv2d_fwd_add) and y += conv(x) (accumulate) of the band kernel: the sum is formed in
    f32 from the unrounded convolution and rounded once — never further from the f32 sum than adding two bf16 tensors is."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    from pmt_learning_for_semantic_segmentation_and_disparity_amd._lib import call, ptr, stream_ptr, dtype_code
    g = torch.Generator().manual_seed(7 * B + ci)
    x = torch.randn(B, H, W, ci, generator=g).cuda().bfloat16().permute(0, 3, 1, 2)
    a = torch.randn(B, H, W, co, generator=g).cuda().bfloat16().permute(0, 3, 1, 2)
    w = (torch.randn(co, ci, 5, 5, generator=g) * 0.05).cuda()
    wp = ops.packed_weight(w, 'conv', 'fwd', torch.bfloat16)
    ref = F.conv2d(x.float().cpu(), w.bfloat16().float().cpu(), None, padding=2) + a.float().cpu()
    y = ops.empty_nhwc(B, co, H, W, torch.bfloat16, "cuda")
    call("sdhip_conv2d_fwd_add", ptr(x), ptr(wp), ptr(y), ptr(a), co, B, H, W, ci, ci, H, W, co, co, 5, 5, 2, 2, dtype_code(x), stream_ptr())
    y2 = a.clone(memory_format=torch.preserve_format)
    ops._conv_launch(x, ci, wp, y2, co, None, None, None, None, B, H, W, ci, H, W, co, 5, 5, 1, 1, 2, 2, False, 1, 0, True)
    torch.cuda.synchronize()
    assert torch.equal(y, y2)
    err = (y.float().cpu() - ref).abs().max().item()
    assert err <= 2.0 ** -8 * ref.abs().max().item()          # one bf16 rounding of the sum


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,k,last,H,W", [(64, 64, 5, False, 128, 256), (32, 32, 3, True, 48, 64), (16, 32, 5, True, 128, 192)])
def test_conv2downup_gradient_slots_match_autograd_adds(cin, cout, k, last, H, W):
    """Conv2DownUp with the skip gradients summed by the data-gradient launches (ops.GradSlot) against the same block with
    every skip gradient returned to autograd: same forward, gradients equal up to the one bf16 rounding the fused sum saves."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N
    torch.manual_seed(3)
    blk = N.Conv2DownUp(cin, cout, k, lastLayer=last).cuda().train()
    x = torch.randn(4, H, W, cin, device="cuda").bfloat16().permute(0, 3, 1, 2)
    gy = torch.randn(4, H, W, cout, device="cuda").bfloat16().permute(0, 3, 1, 2)

    def run(slots):
        old = N.GRAD_SLOTS
        N.GRAD_SLOTS = slots
        try:
            for p in blk.parameters():
                p.grad = None
            xi = x.clone().requires_grad_(True)
            y = blk(xi, groups=2)
            # a consumer that hands the SAME gradient tensor to two inputs (what torch's add does): the block must not write into it
            z = y + y.detach() * 0.0
            z.backward(gy)
            torch.cuda.synchronize()
            return y.detach().float(), xi.grad.float(), [p.grad.float().clone() for p in blk.parameters() if p.grad is not None]
        finally:
            N.GRAD_SLOTS = old

    y1, gx1, gp1 = run(True)
    y0, gx0, gp0 = run(False)
    assert torch.equal(y1, y0)
    assert (gx1 - gx0).abs().max().item() <= 3e-2 * gx0.abs().max().item()
    for a, b in zip(gp1, gp0):
        assert (a - b).abs().max().item() <= 3e-2 * b.abs().max().item() + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("k,B,ci,co,H,W,groups,with_add", [
    (5, 4, 64, 64, 96, 256, 2, False), (5, 4, 64, 64, 100, 250, 2, True), (5, 4, 32, 32, 96, 256, 1, True),
    (3, 4, 64, 64, 48, 64, 2, True), (3, 2, 32, 128, 33, 47, 1, True)])
def test_data_gradient_with_bn_sums_and_addend(k, B, ci, co, H, W, groups, with_add):
    """sdhip_conv2d_fwd_bnbwd on 5x5 and 3x3 shapes (halo-tile kernel with the sums epilogue), with and without a second
    gradient contribution: y = conv (+ addend) equals the separate launches, and the two sums equal sdhip_affine_act_bwd's
    over that y."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    from pmt_learning_for_semantic_segmentation_and_disparity_amd._lib import call, ptr, stream_ptr, dtype_code
    dev = "cuda"
    g = torch.Generator().manual_seed(k * 1000 + H)
    x = torch.randn(B, H, W, ci, generator=g).to(dev).bfloat16().permute(0, 3, 1, 2)
    u = torch.randn(B, H, W, co, generator=g).to(dev).bfloat16().permute(0, 3, 1, 2)
    a = torch.randn(B, H, W, co, generator=g).to(dev).bfloat16().permute(0, 3, 1, 2) if with_add else None
    w = (torch.randn(co, ci, k, k, generator=g) * 0.05).to(dev)
    sc = (torch.rand(groups, co, generator=g) + 0.5).to(dev)
    sh = (torch.randn(groups, co, generator=g) * 0.3).to(dev)
    wp = ops.packed_weight(w, 'conv', 'fwd', torch.bfloat16)
    dt = dtype_code(x)
    pad = k // 2
    # reference: plain launch (+ f32 add of the second tensor, rounded once), then the reduction pass
    y0 = ops.empty_nhwc(B, co, H, W, torch.bfloat16, dev)
    ops._conv_launch(x, ci, wp, y0, co, None, None, None, None, B, H, W, ci, H, W, co, k, k, 1, 1, pad, pad, False, 1, 0, False)
    y1 = ops.empty_nhwc(B, co, H, W, torch.bfloat16, dev)
    sums = torch.zeros(ops.NREP, groups, 2, co, dtype=torch.float64, device=dev)
    call("sdhip_conv2d_fwd_bnbwd", ptr(x), ptr(wp), ptr(y1), ptr(sums), co, ops.NREP, ptr(u), co, ptr(sc), ptr(sh),
         ptr(a) if with_add else None, co if with_add else 0, B, H, W, ci, ci, H, W, co, co, k, k, 1, pad, pad, groups, 0, dt, stream_ptr())
    torch.cuda.synchronize()
    if with_add:
        ref = F.conv2d(x.float().cpu(), w.bfloat16().float().cpu(), None, padding=pad) + a.float().cpu()
        assert (y1.float().cpu() - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item()
    elif k == 5:
        # the plain 5x5 launch runs on the persistent kernel, the one with the sums on the halo-tile kernel: same products,
        # another summation order over (tap, channel half) -> equal up to the bf16 rounding of the stored value
        d = (y0.float() - y1.float()).abs().max().item()
        assert d <= 2.0 ** -7 * y0.float().abs().max().item()
    else:
        assert torch.equal(y0, y1)
    both = torch.zeros(2, ops.NREP, groups, co, dtype=torch.float32, device=dev)
    call("sdhip_affine_act_bwd", ptr(y1), co, ptr(u), co, None, 0, ptr(sc), ptr(sh), ptr(both[0]), ptr(both[1]), ops.NREP,
         B * H * W, co, groups, 1, 0, 0, dt, stream_ptr())
    torch.cuda.synchronize()
    ref_s = both.double().sum(1)                  # [2][groups][C]
    got = sums.sum(0).permute(1, 0, 2)
    assert (ref_s - got).abs().max().item() <= 3e-4 * ref_s.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("k,dil,B,co,H,W", [(5, 1, 2, 64, 100, 450), (1, 1, 4, 64, 33, 47), (3, 2, 2, 32, 40, 70), (5, 1, 2, 24, 16, 32)])
def test_fanout_conv_one_input_channel(k, dil, B, co, H, W):
    """conv_fanout_kernel (one input channel -> <= 64 output channels, taps as the MFMA reduction axis; the data gradient of
    the disparity head and of the attention gates) against an f32 convolution of the same bf16-rounded operands."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    g = torch.Generator().manual_seed(k * 31 + co)
    x = torch.randn(B, H, W, 1, generator=g).cuda().bfloat16()
    xp, ldx = ops.alloc_nhwc(B, 1, H, W, torch.bfloat16, "cuda")            # pixel stride padded to 8 elements, as in the step
    xp.copy_(x.permute(0, 3, 1, 2))
    w = (torch.randn(co, 1, k, k, generator=g) * 0.2).cuda()
    wp = ops.packed_weight(w, 'conv', 'fwd', torch.bfloat16)
    pad = dil * (k - 1) // 2
    y = ops.empty_nhwc(B, co, H, W, torch.bfloat16, "cuda")
    ops._conv_launch(xp, ldx, wp, y, co, None, None, None, None, B, H, W, 1, H, W, co, k, k, 1, dil, pad, pad, False, 1, 0, False)
    torch.cuda.synchronize()
    ref = F.conv2d(x.permute(0, 3, 1, 2).float().cpu(), w.bfloat16().float().cpu(), None, padding=pad, dilation=dil)
    assert (y.float().cpu() - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item() + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("k,B,ci,H,W,act", [(5, 2, 64, 100, 450, 0), (3, 4, 32, 33, 47, 1), (5, 2, 24, 16, 32, 2), (1, 2, 64, 40, 70, 0), (3, 1, 16, 8, 31, 0)])
def test_fanin_conv_one_output_map(k, B, ci, H, W, act):
    """conv_fanin_kernel (16..64 input channels -> ONE output map: taps as the MFMA rows, P[tap][halo pixel] through LDS, a
    fixed-order sum over taps; the single-map heads of the 2-D networks) against an f32 convolution of the same bf16-rounded
    operands, with bias and activation; two launches agree bit for bit."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    g = torch.Generator().manual_seed(k * 31 + ci)
    x = torch.randn(B, ci, H, W, generator=g).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
    xv, ldx = ops.nhwc_view(x)
    w = (torch.randn(1, ci, k, k, generator=g) * 0.1).cuda()
    bias = torch.full((1,), 0.25, device="cuda")
    wp = ops.packed_weight(w, 'conv', 'fwd', torch.bfloat16)
    pad = k // 2
    ys = []
    for _ in range(2):
        y, ldy = ops.alloc_nhwc(B, 1, H, W, torch.bfloat16, "cuda")
        ops._conv_launch(xv, ldx, wp, y, ldy, bias, None, None, None, B, H, W, ci, H, W, 1, k, k, 1, 1, pad, pad, False, 1, act, False)
        ys.append(y)
    torch.cuda.synchronize()
    ref = F.conv2d(x.float().cpu(), w.bfloat16().float().cpu(), bias.cpu(), padding=pad)
    ref = torch.relu(ref) if act == 1 else (torch.sigmoid(ref) if act == 2 else ref)
    assert torch.equal(ys[0], ys[1])
    assert (ys[0].float().cpu() - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item() + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("k,ci,co", [(5, 64, 64), (5, 32, 32), (3, 32, 64), (3, 32, 32), (3, 16, 24)])
def test_band_full_size_translation_equivariance(k, ci, co):
    """Size-independent property at the benchmark's full size (8 x 256 x 512): shifting the input by one tile (16 rows, 32
    columns) shifts the output by the same amount, bit for bit, wherever the receptive field stays inside the image — every
    tile boundary, halo row and persistent-workgroup hand-over of the band kernel is exercised, with no reference needed."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    B, H, W = 8, 256, 512
    g = torch.Generator(device="cuda").manual_seed(k * 100 + ci)
    x = torch.randn(B, H, W, ci, device="cuda", generator=g).bfloat16().permute(0, 3, 1, 2)
    w = torch.randn(co, ci, k, k, device="cuda", generator=g) * 0.05
    xs = torch.zeros_like(x)
    xs[:, :, 16:, 32:] = x[:, :, :-16, :-32]
    y = ops.conv2d(x, w, None, padding='same')
    ys = ops.conv2d(xs, w, None, padding='same')
    torch.cuda.synchronize()
    p = k // 2            # rows / columns shifted in from outside are zeros in both (padding there, the cleared border here);
    a = ys[:, :, 16:H - p, 32:W - p]      # only outputs whose window reaches past the far border of the SHIFTED image differ
    b = y[:, :, :H - 16 - p, :W - 32 - p]
    assert torch.equal(a, b)
    assert float(y.float().abs().max()) > 1.0 and bool(torch.isfinite(y.float()).all())
