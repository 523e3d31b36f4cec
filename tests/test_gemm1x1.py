"""1x1 convolutions on the streaming-GEMM kernel (csrc/conv_gemm.h) against an f32 contraction of the same bf16 operands:
plain, with the BatchNorm+ReLU prologue per statistics group, with the statistics epilogue, odd channel counts (masked
tail), bias + activation, and the gathered forms (channel concat of two tensors, nearest-neighbour upsampled segment) —
models/densenet.py:41-45,119-128 and the concat -> 1x1 -> ReLU sites of models/dsnet_t2.py:1206-1216."""
import pytest
import torch
import torch.nn.functional as F


def _rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12))


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,Cin,ldx,Cout,pro,groups,stats", [
    (16, 64, 64, 64, 64, 128, True, 2, True),       # DenseNet conv1, first layer of a block
    (16, 64, 64, 224, 256, 128, True, 2, True),     # channel prefix of a wider slab, partial last chunk
    (4, 64, 128, 128, 128, 224, False, 1, False),   # data gradient of a bottleneck: two cout blocks
    (8, 64, 128, 1024, 1024, 512, True, 2, False),  # transition-like: 16 chunks, 4 cout blocks
    (6, 64, 96, 65, 72, 64, False, 1, False),       # odd channel count in a padded pixel stride
    (6, 64, 96, 64, 64, 65, False, 1, False),       # odd output channels
    (8, 48, 88, 32, 32, 33, False, 1, True),        # ragged last tile (pixels not a multiple of 256) + statistics
    (16, 8, 16, 992, 1024, 128, True, 2, True)])    # small map: stays on the halo-tile kernel (same contract)
def test_gemm1x1_matches_f32_contraction(B, H, W, Cin, ldx, Cout, pro, groups, stats):
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    torch.manual_seed(Cin * 7 + Cout)
    dev = torch.device("cuda:0")
    slab = torch.randn(B, H, W, ldx, device=dev).to(torch.bfloat16)
    if ldx != Cin:
        slab[..., Cin:] = float("nan")               # pad / foreign channels must never reach the result
    x = slab.permute(0, 3, 1, 2)[:, :Cin]
    w = torch.randn(Cout, Cin, 1, 1, device=dev) * (1.0 / Cin ** 0.5)
    wp = ops.packed_weight(w, 'conv', 'fwd', torch.bfloat16)
    y, ldy = ops.alloc_nhwc(B, Cout, H, W, torch.bfloat16, dev)
    sc = (torch.rand(groups, Cin, device=dev) + 0.5) if pro else None
    sh = (torch.rand(groups, Cin, device=dev) - 0.5) if pro else None
    st = torch.zeros(ops.NREP, groups, 2, Cout, dtype=torch.float64, device=dev) if stats else None
    ops._conv_launch(x, ldx, wp, y, ldy, None, sc, sh, st, B, H, W, Cin, H, W, Cout, 1, 1, 1, 1, 0, 0, pro, groups, 0, False, ops.NREP)
    xe = x.float()
    if pro:
        per = B // groups
        xe = torch.relu(torch.addcmul(sh.repeat_interleave(per, 0)[:, :, None, None], xe, sc.repeat_interleave(per, 0)[:, :, None, None]))
        xe = xe.to(torch.bfloat16).float()           # the kernel rounds the prologue result to bf16
    want = F.conv2d(xe, w.to(torch.bfloat16).float())
    assert torch.isfinite(y.float()).all()
    assert _rel(y, want) < 6e-3
    if stats:
        got = st.sum(0)                              # [groups][2][Cout]
        yq = y.float().view(groups, B // groups, Cout, H * W)
        assert torch.allclose(got[:, 0].float(), yq.sum((1, 3)), rtol=2e-3, atol=2e-2)
        assert torch.allclose(got[:, 1].float(), (yq * yq).sum((1, 3)), rtol=2e-3, atol=2e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,c0,us0,c1,us1,Cout,act,bias", [
    (2, 64, 128, 64, 3, 1, 0, 64, 1, False),         # conv1d_2: cat(x8 nearest upsample of a 64-channel map, 1-channel map) -> 64, ReLU
    (2, 64, 128, 256, 0, 256, 0, 128, 1, False),     # conv1d_4: cat of the two towers' pyramids
    (2, 32, 64, 32, 1, 1, 0, 32, 1, True),           # segNet.conv1d_2 with a bias
    (2, 64, 64, 64, 0, 0, 0, 1, 2, True)])           # single segment, one output channel, sigmoid (attention maps)
def test_conv1x1_cat_matches_torch(B, H, W, c0, us0, c1, us1, Cout, act, bias):
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    torch.manual_seed(c0 + c1 + Cout)
    dev = torch.device("cuda:0")
    a, _ = ops.alloc_nhwc(B, c0, H >> us0, W >> us0, torch.bfloat16, dev)
    a.copy_(torch.randn(B, c0, H >> us0, W >> us0, device=dev))
    segs = [(a, us0)]
    full = [F.interpolate(a.float(), scale_factor=1 << us0, mode='nearest') if us0 else a.float()]
    if c1:
        b, _ = ops.alloc_nhwc(B, c1, H >> us1, W >> us1, torch.bfloat16, dev)
        b.copy_(torch.randn(B, c1, H >> us1, W >> us1, device=dev))
        segs.append((b, us1))
        full.append(F.interpolate(b.float(), scale_factor=1 << us1, mode='nearest') if us1 else b.float())
    w = torch.randn(Cout, c0 + c1, 1, 1, device=dev) * (1.0 / (c0 + c1) ** 0.5)
    bv = torch.randn(Cout, device=dev) if bias else None
    y = ops.conv1x1_cat_forward(segs, w, bv, act)
    want = F.conv2d(torch.cat(full, 1), w.to(torch.bfloat16).float(), bv)
    want = torch.relu(want) if act == 1 else (torch.sigmoid(want) if act == 2 else want)
    assert tuple(y.shape) == tuple(want.shape) and torch.isfinite(y.float()).all()
    assert _rel(y, want) < 6e-3


@pytest.mark.gpu
@pytest.mark.parametrize("B,h,w,us,c0,c1,Cout", [(2, 32, 64, 3, 64, 1, 64), (2, 64, 128, 2, 64, 1, 32), (2, 16, 32, 4, 32, 1, 32)])
def test_upcat_conv1x1_forward_backward_match_torch(B, h, w, us, c0, c1, Cout):
    """The fused upsample + concat + 1x1 + ReLU node (ops.upcat_conv1x1) against torch's interpolate / cat / conv2d / relu on
    the same bf16-rounded operands: output, both input gradients and the weight gradient (whose two halves are computed
    without ever forming the upsampled map)."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    torch.manual_seed(us * 100 + c0)
    dev = torch.device("cuda:0")
    H, W = h << us, w << us
    xs0 = torch.randn(B, c0, h, w, device=dev).to(torch.bfloat16)
    xf0 = torch.randn(B, c1, H, W, device=dev).to(torch.bfloat16)
    w0 = (torch.randn(Cout, c0 + c1, 1, 1, device=dev) * (1.0 / (c0 + c1) ** 0.5))
    gy = (torch.randn(B, Cout, H, W, device=dev) * 0.1).to(torch.bfloat16)
    xs, _ = ops.alloc_nhwc(B, c0, h, w, torch.bfloat16, dev); xs.copy_(xs0); xs.requires_grad_(True)
    xf, _ = ops.alloc_nhwc(B, c1, H, W, torch.bfloat16, dev); xf.copy_(xf0); xf.requires_grad_(True)
    wt = w0.clone().requires_grad_(True)
    ops.set_step_context(None)
    y = ops.upcat_conv1x1(xs, xf, wt, act=1)
    assert y is not None
    y.backward(gy)
    rs, rf = xs0.float().requires_grad_(True), xf0.float().requires_grad_(True)
    rw = w0.to(torch.bfloat16).float().requires_grad_(True)
    want = torch.relu(F.conv2d(torch.cat([F.interpolate(rs, scale_factor=1 << us, mode='nearest'), rf], 1), rw))
    want.backward(gy.float() * (y.float() > 0))          # same ReLU mask as the bf16 output (values at the rounding edge)
    assert _rel(y, want) < 6e-3
    assert _rel(xs.grad, rs.grad) < 2e-2, _rel(xs.grad, rs.grad)
    assert _rel(xf.grad, rf.grad) < 2e-2, _rel(xf.grad, rf.grad)
    assert _rel(wt.grad, rw.grad) < 2e-2, _rel(wt.grad, rw.grad)
