"""Correlation sampler: oracle self-consistency (CPU) and HIP-vs-oracle parity (GPU)."""
import numpy as np
import pytest
import torch

from oracle import cbuild
from oracle.detweights import randn_input
from oracle.ref_models import SpatialCorrelationSampler as RefCorr

CASES = [  # B, C, H, W, patch, dil_patch
    (2, 16, 6, 9, (1, 17), 1),
    (1, 352, 8, 16, (1, 17), 1),    # the shipped 1-D configuration's channel count
    (2, 24, 7, 10, (17, 17), 1),    # 2-D, image smaller than the patch
    (1, 5, 4, 6, (3, 5), 2),        # ragged channels (scalar path), dilated patch
    (1, 8, 1, 1, (1, 17), 1),       # single pixel: everything but the centre is out of range
]


@pytest.mark.parametrize("B,C,H,W,patch,dil", CASES)
def test_oracle_torch_matches_c(oracle_clib, B, C, H, W, patch, dil):
    a = randn_input(1, "a", (B, C, H, W)).requires_grad_(True)
    b = randn_input(1, "b", (B, C, H, W)).requires_grad_(True)
    y = RefCorr(1, patch, 1, 0, 1, dil)(a, b)
    yc = cbuild.corr_forward(oracle_clib, a.detach().numpy(), b.detach().numpy(), patch, dil)
    np.testing.assert_allclose(y.detach().numpy(), yc, rtol=1e-4, atol=1e-4)
    g = randn_input(2, "g", tuple(y.shape))
    y.backward(g)
    g1, g2 = cbuild.corr_backward(oracle_clib, a.detach().numpy(), b.detach().numpy(), g.numpy(), patch, dil)
    np.testing.assert_allclose(a.grad.numpy(), g1, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(b.grad.numpy(), g2, rtol=1e-4, atol=1e-4)


def test_oracle_known_answer(oracle_clib):
    """Hand-checkable vector: displacement index j pairs in1[w] with in2[w + j - PW//2], zero outside."""
    a = np.zeros((1, 1, 1, 4), np.float32); b = np.zeros((1, 1, 1, 4), np.float32)
    a[0, 0, 0] = [1, 2, 3, 4]; b[0, 0, 0] = [10, 20, 30, 40]
    y = cbuild.corr_forward(oracle_clib, a, b, (1, 3))[0, 0]       # (PW, H, W)
    assert y[0, 0].tolist() == [0, 2 * 10, 3 * 20, 4 * 30]       # j=0: in2 shifted by -1
    assert y[1, 0].tolist() == [10, 40, 90, 160]                  # j=1: aligned
    assert y[2, 0].tolist() == [1 * 20, 2 * 30, 3 * 40, 0]        # j=2: in2 shifted by +1


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("B,C,H,W,patch,dil", CASES + [
    (2, 352, 32, 64, (1, 17), 1), (2, 256, 16, 32, (1, 17), 1),
    # the tiled MFMA kernels (bf16): the 2-D correlation of dsnet (models/dsnet_t2.py:129-133,221-223) at its bench-shaped
    # map, ragged maps, a channel count that is not a multiple of the 64-channel chunk, the (1, 21) patch
    (2, 352, 32, 64, (17, 17), 1), (1, 64, 13, 21, (17, 17), 1), (2, 40, 9, 30, (1, 21), 1), (1, 448, 16, 24, (17, 17), 1)])
def test_hip_matches_oracle(oracle_clib, B, C, H, W, patch, dil, dtype, tol):
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.nn import SpatialCorrelationSampler
    a = randn_input(1, "a", (B, C, H, W)); b = randn_input(1, "b", (B, C, H, W))
    if dtype == torch.bfloat16:  # the oracle sees the same (rounded) inputs
        a = a.bfloat16().float(); b = b.bfloat16().float()
    yc = cbuild.corr_forward(oracle_clib, a.numpy(), b.numpy(), patch, dil)
    g = randn_input(2, "g", yc.shape)
    if dtype == torch.bfloat16:
        g = g.bfloat16().float()
    g1, g2 = cbuild.corr_backward(oracle_clib, a.numpy(), b.numpy(), g.numpy(), patch, dil)

    ad = a.cuda().to(dtype).requires_grad_(True); bd = b.cuda().to(dtype).requires_grad_(True)
    y = SpatialCorrelationSampler(1, patch, 1, 0, 1, dil)(ad, bd)
    assert tuple(y.shape) == yc.shape
    y.backward(g.cuda().to(dtype))
    scale = max(1.0, float(np.abs(yc).max()))
    assert float((y.float().cpu() - torch.from_numpy(yc)).abs().max()) <= tol * scale
    for got, want in ((ad.grad, g1), (bd.grad, g2)):
        s = max(1.0, float(np.abs(want).max()))
        assert float((got.float().cpu() - torch.from_numpy(want)).abs().max()) <= tol * s


@pytest.mark.gpu
def test_hip_corr_nchw_contiguous_input():
    """The reference hands the op NCHW-contiguous tensors; the drop-in accepts them unchanged."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.nn import SpatialCorrelationSampler
    a = randn_input(3, "a", (2, 32, 8, 12)); b = randn_input(3, "b", (2, 32, 8, 12))
    want = RefCorr(1, (1, 17))(a, b)
    got = SpatialCorrelationSampler(1, (1, 17), 1, 0, 1, 1)(a.cuda(), b.cuda())
    assert torch.squeeze(got, 1).shape == (2, 17, 8, 12)
    assert float((got.cpu() - want).abs().max()) < 1e-4
