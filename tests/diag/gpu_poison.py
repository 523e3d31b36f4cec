"""Uninitialised-memory hunt: every torch.empty / empty_like on the GPU is filled with NaN, then forward + backward of the
networks must still produce finite outputs and gradients.  Prints the first offenders."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N, ops
from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import synthetic_batch

_empty, _empty_like = torch.empty, torch.empty_like
def _poison(t):
    if t.is_cuda and t.is_floating_point():
        t.fill_(float("nan"))
    elif t.is_cuda and t.dtype in (torch.int32, torch.int64, torch.uint8):
        t.fill_(0x7f if t.dtype == torch.uint8 else 0x7fffff00)
    return t
torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))

def pos(B, H, W):
    h = (torch.arange(0, H) * 1024 // H).unsqueeze(0).unsqueeze(2).expand(B, -1, W) // 8
    w = (torch.arange(0, W) * 2048 // W).unsqueeze(0).unsqueeze(1).expand(B, H, -1) // 16
    return h.cuda(), w.cuda()

def check(tag, m, args, loss_of, dt):
    outs = m(*args)
    bad_o = [i for i, o in enumerate(outs) if torch.is_tensor(o) and not torch.isfinite(o.float()).all()]
    loss = loss_of(outs)
    loss.backward()
    bad = [k for k, p in m.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print("%-34s %s loss %.4f  nonfinite outputs %s  nonfinite grads %d %s" % (tag, str(dt).split('.')[-1], float(loss), bad_o, len(bad), bad[:5]), flush=True)

which = sys.argv[1:] or ["hanet", "mini", "dsnet", "psm"]
for dt in (torch.float32, torch.bfloat16):
    B, H, W = 2, 256, 256
    if "hanet" in which:
        for aspp in (0, 2):
            torch.manual_seed(0)
            m = N.minidsnetExt(N.CFG(aspp=aspp, hanet=1), labels=19, patch_type='1dcorr').cuda().train()
            l, r, seg, disp = synthetic_batch(B, H, W, labels=19, seed=7)
            check("mini aspp=%d hanet=1 labels=19" % aspp, m, (l.to(dt), r.to(dt), pos(B, H, W)),
                  lambda o: ops.train_loss(o[0], o[1], o[2], seg, disp, True, True), dt)
    if "mini" in which:
        torch.manual_seed(0)
        m = N.minidsnetExt(N.CFG(), labels=2, patch_type='1dcorr').cuda().train()
        l, r, seg, disp = synthetic_batch(B, H, W, seed=7)
        check("mini aspp=0 labels=2", m, (l.to(dt), r.to(dt)), lambda o: ops.train_loss(o[0], o[1], o[2], seg, disp, True), dt)
        m = N.minidsnetExt(N.CFG(aspp=1, dropout=0.2), labels=2, patch_type='').cuda().train()
        check("mini aspp=1 2dcorr dropout=.2", m, (l.to(dt), r.to(dt)), lambda o: ops.train_loss(o[0], o[1], o[2], seg, disp, True), dt)
    if "dsnet" in which:
        torch.manual_seed(0)
        l, r, seg, disp = synthetic_batch(B, H, W, seed=7)
        for cls in (N.dsnet, N.dsnetnoCorr):
            m = cls(N.CFG(), labels=2).cuda().train()
            check(cls.__name__, m, (l.to(dt), r.to(dt)), lambda o: ops.train_loss(o[0], o[1], o[2], seg, disp, True), dt)
    if "psm" in which:
        from pmt_learning_for_semantic_segmentation_and_disparity_amd.psmnet import PSMNet
        torch.manual_seed(0)
        l, r, seg, disp = synthetic_batch(B, H, W, seed=7)
        m = PSMNet(64).cuda().train()
        check("PSMNet(64)", m, (l.to(dt), r.to(dt)), lambda o: ops.mean_l1_loss(o, disp[:, 0]), dt)

# the training step itself (zero arena, queued + grouped weight gradients, fused Adam), eager and captured
if "step" in (sys.argv[1:] or ["step"]):
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.psmnet import PSMNet
    for name, mk, lf in (("minidsnetExt", lambda: N.minidsnetExt(N.CFG(), labels=2, patch_type='1dcorr').cuda().train(), None),
                         ("dsnet", lambda: N.dsnet(N.CFG(), labels=2).cuda().train(),
                          lambda o, seg, disp: ops.train_loss(o[0], o[1], o[2], seg, disp, True)),
                         ("PSMNet(64)", lambda: PSMNet(64).cuda().train(), lambda o, seg, disp: ops.mean_l1_loss(o, disp[:, 0]))):
        for graph in (False, True):
            torch.manual_seed(0)
            batch = synthetic_batch(2, 256, 256, seed=7)
            ts = TrainStep(mk(), dtype=torch.bfloat16, use_graph=graph, lr=1e-4, loss_fn=lf)
            losses = [float(ts(*batch)) for _ in range(3)]
            ok = all(l == l for l in losses) and bool(torch.isfinite(ts.flat_g).all()) and bool(torch.isfinite(ts.flat_p).all())
            print("%-14s TrainStep graph=%-5s losses %s  finite gradients / parameters: %s" % (name, graph, ["%.4f" % l for l in losses], ok), flush=True)
            ops.set_step_context(None)
            del ts
