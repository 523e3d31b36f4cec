"""Round-2 diagnostics: (1) where the NaN of the 19-class HANet bf16 backward comes from, (2) bf16 vs f32 per output and
the f32 path's own sensitivity to a tiny input perturbation, (3) the checkpoint/graph loss mystery."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from oracle import ref_models as R
from oracle.detweights import fill_state_dict
from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N, ops, checkpoint as C
from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch

what = sys.argv[1:] or ["nan", "bf16", "ckpt"]

def pos(B, H, W):
    h = (torch.arange(0, H) * 1024 // H).unsqueeze(0).unsqueeze(2).expand(B, -1, W) // 8
    w = (torch.arange(0, W) * 2048 // W).unsqueeze(0).unsqueeze(1).expand(B, H, -1) // 16
    return h.cuda(), w.cuda()

if "nan" in what and False:
    for (B, H, W, dt) in ((2, 256, 256, torch.bfloat16), (2, 512, 1024, torch.bfloat16), (2, 512, 1024, torch.float32)):
        torch.manual_seed(0)
        m = N.minidsnetExt(N.CFG(aspp=0, hanet=1), labels=19, patch_type='1dcorr').cuda().train()
        left, right, seg, disp = synthetic_batch(B, H, W, labels=19, seed=7)
        outs = m(left.to(dt), right.to(dt), pos(B, H, W))
        for o, n in zip(outs[:3], ("seg1", "disp", "seg2")):
            o.retain_grad()
        loss = ops.train_loss(outs[0], outs[1], outs[2], seg, disp, True, True)
        loss.backward()
        bad = [k for k, p in m.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        print("NAN", B, H, W, dt, "loss", float(loss), "nonfinite grads:", len(bad), bad[:6], bad[-3:], flush=True)
        for o, n in zip(outs[:3], ("seg1", "disp", "seg2")):
            print("   d%s finite=%s absmax=%.3e" % (n, bool(torch.isfinite(o.grad.float()).all()), float(o.grad.float().abs().max())), flush=True)

if "bf16" in what:
    def rel(a, b):
        return float((a.float() - b.float()).norm() / b.float().norm())
    for (B, H, W) in ((2, 256, 256), (8, 256, 512)):
        left, right, seg, disp = synthetic_batch(B, H, W, seed=3)
        m = fill_state_dict(N.minidsnetExt(R.CFG(aspp=0), labels=2, patch_type='1dcorr'), 31).cuda().train()
        with torch.no_grad():
            o32 = m(left, right)
            t32 = m.resnet_features(torch.cat([left, right]), groups=2)
        m = fill_state_dict(N.minidsnetExt(R.CFG(aspp=0), labels=2, patch_type='1dcorr'), 31).cuda().train()
        with torch.no_grad():
            noise = 1e-4
            o32p = m(left * (1 + noise * torch.randn_like(left)), right * (1 + noise * torch.randn_like(right)))
        m = fill_state_dict(N.minidsnetExt(R.CFG(aspp=0), labels=2, patch_type='1dcorr'), 31).cuda().train()
        with torch.no_grad():
            o16 = m(left.bfloat16(), right.bfloat16())
            t16 = m.resnet_features(torch.cat([left, right]).bfloat16(), groups=2)
        print("BF16 B=%d %dx%d" % (B, H, W))
        for i, n in enumerate(("seg1", "disp", "seg2")):
            print("   %-5s bf16-vs-f32 relL2 %.4f | f32 response to 1e-4 relative input noise %.5f  (amplification %.0fx)" % (
                n, rel(o16[i], o32[i]), rel(o32p[i], o32[i]), rel(o32p[i], o32[i]) / 1e-4), flush=True)
        for i in range(len(t32)):
            print("   tap/pyramid %d  C=%d  bf16-vs-f32 relL2 %.4f" % (i, t32[i].shape[1], rel(t16[i], t32[i])), flush=True)
    # eval mode with trained-like running statistics (tests/golden/cfg5.npz): bf16 vs the f32 reference fixture
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from test_parity_r2 import _cfg5_case, GDIR
    from test_nets import _sample
    gold = np.load(os.path.join(GDIR, "cfg5.npz"))
    for tag in ("a2_hanet_l19", "a0_hanet_l19"):
        m, a, b, pos_, seg, disp = _cfg5_case(gold, tag, N.minidsnetExt, "cuda")
        with torch.no_grad():
            outs = m(a.bfloat16(), b.bfloat16(), pos_)
        for i, n in enumerate(("seg1", "disp", "seg2")):
            want = gold["%s.eval.%s.sample" % (tag, n)]
            got = _sample(outs[i], 8)
            print("EVAL %s %-5s bf16-vs-f32-golden relL2 %.4f" % (tag, n, float(np.linalg.norm(got - want) / np.linalg.norm(want))), flush=True)
    # loss trajectories
    batch = synthetic_batch(2, 256, 256, seed=5)
    for dt in (torch.float32, torch.bfloat16):
        m = fill_state_dict(N.minidsnetExt(R.CFG(aspp=0), labels=2, patch_type='1dcorr'), 31).cuda().train()
        ts = TrainStep(m, dtype=dt, use_graph=True)
        tr = [float(ts(*batch)) for _ in range(48)]
        ops.set_step_context(None)
        print("TRAJ", dt, " ".join("%.3f" % v for v in tr), flush=True)

if "ckpt" in what:
    batch = synthetic_batch(2, 256, 256)
    mk = lambda seed: fill_state_dict(N.minidsnetExt(R.CFG(), labels=2, patch_type='1dcorr'), seed).cuda().train()
    a = TrainStep(mk(5), dtype=torch.float32, use_graph=True)
    print("CKPT a:", [round(float(a(*batch)), 4) for _ in range(3)], flush=True)
    p5 = a.flat_p.clone()
    state = C.make_state(a, epoch=1)
    C.save_checkpoint(state, 0.0, 0.5, 1.0, 0.5, "/tmp/ckg")
    print("   flat_p unchanged by save:", bool(torch.equal(p5, a.flat_p)), flush=True)
    b = TrainStep(mk(6), dtype=torch.float32, use_graph=False)
    print("   flat_p(a) unchanged by creating b:", bool(torch.equal(p5, a.flat_p)), flush=True)
    C.load_checkpoint_and_params("/tmp/ckg.pth.tar", b, map_location="cuda:0")
    print("   flat_p(a) unchanged by loading b:", bool(torch.equal(p5, a.flat_p)), " b == a:", bool(torch.equal(b.flat_p, a.flat_p)), flush=True)
    la = float(a(*batch))
    print("   la (replay 6) = %.4f" % la, flush=True)
    lb = float(b(*batch))
    print("   lb (eager from checkpoint) = %.4f" % lb, flush=True)
    ops.set_step_context(None)
    with torch.no_grad():
        o = b.model(batch[0], batch[1])
    print("   b fresh forward: %.4f" % float(ops.train_loss(o[0], o[1], o[2], batch[2], batch[3], True)), flush=True)
