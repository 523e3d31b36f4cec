"""One tiny hot-path invocation on cuda:0, checked against the CPU oracle (imported here only as the checker)."""
import torch


def run():
    from oracle.ref_models import SpatialCorrelationSampler as RefCorr
    from oracle.detweights import randn_input
    from .nn import SpatialCorrelationSampler
    a = randn_input(5, "a", (2, 64, 8, 16)); b = randn_input(5, "b", (2, 64, 8, 16))
    ar = a.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    want = RefCorr(1, (1, 17))(ar, br)
    want.sum().backward()
    ad = a.cuda().requires_grad_(True); bd = b.cuda().requires_grad_(True)
    got = SpatialCorrelationSampler(1, (1, 17), 1, 0, 1, 1)(ad, bd)
    got.sum().backward()
    torch.cuda.synchronize()
    err = float((got.cpu() - want).abs().max())
    gerr = float((ad.grad.cpu() - ar.grad).abs().max())
    assert err < 1e-3 and gerr < 1e-3, (err, gerr)
    print("smoke ok: corr fwd err %.2e, bwd err %.2e" % (err, gerr))
