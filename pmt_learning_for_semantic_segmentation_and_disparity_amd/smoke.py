"""One small hot-path invocation on cuda:0, checked against the CPU oracle (imported here only as the checker):
the correlation operator alone, then ONE training step (forward + loss + backward) of `minidsnetExt` at 2 x 256 x 256 in
fp32 — conv / BatchNorm / pooling / resize / correlation / loss kernels of libsdhip vs oracle/ref_models.py."""
import torch
import torch.nn.functional as F


def _corr():
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
    return err, gerr


def _step():
    from oracle import ref_models as R
    from oracle.detweights import fill_state_dict, rand_input
    from . import nn as N
    cfg = R.CFG(aspp=0)
    H = W = 256   # smallest square the 128-pixel pyramid pool of piramidNet2 accepts (models/dsnet_t2.py:420-470)
    a, b = rand_input(7, "left", (2, 3, H, W)), rand_input(7, "right", (2, 3, H, W))
    seg = F.one_hot((rand_input(7, "seg", (2, H, W)) > 0.5).long(), 2).permute(0, 3, 1, 2).float()
    disp = rand_input(7, "disp", (2, 1, H, W), 0.0, 8.0)

    def loss_of(outs, seg, disp):
        ce = lambda y: torch.mean(torch.sum(-seg * F.log_softmax(y.float(), 1), 1))
        return ce(outs[0]) + ce(outs[2]) + F.l1_loss(outs[1].float(), disp)

    ref = fill_state_dict(R.minidsnetExt(cfg, labels=2, patch_type='1dcorr'), 9).train()
    lo = loss_of(ref(a, b), seg, disp)
    lo.backward()
    m = fill_state_dict(N.minidsnetExt(cfg, labels=2, patch_type='1dcorr'), 9).cuda().train()
    outs = m(a.cuda(), b.cuda())
    lg = loss_of(outs, seg.cuda(), disp.cuda())
    lg.backward()
    torch.cuda.synchronize()
    lerr = abs(float(lg) - float(lo))
    gn = lambda mod: float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in mod.parameters() if p.grad is not None)))
    g0, g1 = gn(ref), gn(m)
    assert lerr < 1e-3 * max(1.0, abs(float(lo))), (float(lg), float(lo))
    assert abs(g1 - g0) < 2e-2 * max(1.0, g0), (g1, g0)
    return lerr, abs(g1 - g0) / max(1e-12, g0)


def run():
    err, gerr = _corr()
    lerr, grel = _step()
    print("smoke ok: corr fwd err %.2e bwd err %.2e; minidsnetExt step loss err %.2e, grad-norm rel err %.2e" % (err, gerr, lerr, grel))
