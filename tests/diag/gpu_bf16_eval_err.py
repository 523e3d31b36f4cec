"""bf16 HIP vs the f32 reference goldens in EVAL mode (no batch-statistics chain): relative L2 per head."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.nn.functional as F
from oracle import ref_models as R
from oracle.detweights import fill_state_dict, rand_input
from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N
from pmt_learning_for_semantic_segmentation_and_disparity_amd.psmnet import PSMNet
import test_nets as TN, test_parity_r2 as TP, test_psmnet as TPS
G = TN.GDIR

def rel(gold, key, t, stride=8):
    want = gold[key + ".sample"]; got = TN._sample(t, stride)
    return float(np.linalg.norm(got - want) / max(1e-12, np.linalg.norm(want))), float(np.abs(got - want).max()), float(np.abs(want).max())

gold = np.load(os.path.join(G, "nets.npz"))
a, b, seg, disp = TN._net_inputs()
for tag, patch, aspp in (("mini_a0", "1dcorr", 0), ("mini_a0_2d", "", 0), ("mini_a1", "1dcorr", 1), ("mini_a2", "1dcorr", 2)):
    for dt in (torch.float32, torch.bfloat16):
        m = fill_state_dict(N.minidsnetExt(R.CFG(aspp=aspp), labels=2, patch_type=patch), 31).cuda().eval()
        with torch.no_grad():
            outs = m(a.cuda().to(dt), b.cuda().to(dt))
        print(tag, dt, [("%.4f" % rel(gold, "%s.eval.%s" % (tag, n), outs[i])[0]) for i, n in enumerate(("seg1", "disp", "seg2"))], flush=True)
gold = np.load(os.path.join(G, "dsnet.npz"))
for dt in (torch.float32, torch.bfloat16):
    m = fill_state_dict(N.dsnet(R.CFG(), labels=2), 61).cuda().eval()
    x, y = rand_input(61, "left", (2, 3, 256, 256)).cuda(), rand_input(61, "right", (2, 3, 256, 256)).cuda()
    with torch.no_grad():
        outs = m(x.to(dt), y.to(dt))
    print("dsnet", dt, [("%.4f" % rel(gold, "dsnet.eval.%s" % n, outs[i])[0]) for i, n in enumerate(("seg1", "disp", "seg2", "disp2"))], flush=True)
gold = np.load(os.path.join(G, "cfg5.npz"))
for tag in ("a2_hanet_l19", "a0_hanet_l19"):
    for dt in (torch.float32, torch.bfloat16):
        m, x, y, pos, seg5, disp5 = TP._cfg5_case(gold, tag, N.minidsnetExt, "cuda")
        with torch.no_grad():
            outs = m(x.to(dt), y.to(dt), pos)
        print(tag, dt, [("%.4f" % rel(gold, "%s.eval.%s" % (tag, n), outs[i])[0]) for i, n in enumerate(("seg1", "disp", "seg2"))], flush=True)
gold = np.load(os.path.join(G, "psmnet.npz"))
for dt in (torch.float32, torch.bfloat16):
    m = TPS._load_eval_stats(fill_state_dict(PSMNet(64), 41), gold).cuda().eval()
    x, y = rand_input(41, "left", (2, 3, 256, 256)).cuda(), rand_input(41, "right", (2, 3, 256, 256)).cuda()
    with torch.no_grad():
        o = m(x.to(dt), y.to(dt))
    o = o[0] if isinstance(o, tuple) else o
    want = gold["psm64.eval.pred0.sample"]; got = o.float().cpu()[:, ::8, ::8].numpy()
    print("psm64", dt, "rel %.4f maxabs %.4f of %.2f" % (np.linalg.norm(got - want) / np.linalg.norm(want), np.abs(got - want).max(), np.abs(want).max()), flush=True)

# ---- gradient norms per top-level module, bf16 eval vs golden (f32 reference)
def gn(m):
    acc = {}
    for k, q in m.named_parameters():
        if q.grad is not None:
            acc[k.split(".")[0]] = acc.get(k.split(".")[0], 0.0) + float(q.grad.double().pow(2).sum())
    return {k: v ** 0.5 for k, v in acc.items()}
gold = np.load(os.path.join(G, "nets.npz"))
for tag, patch in (("mini_a0", "1dcorr"), ("mini_a0_2d", "")):
    m = fill_state_dict(N.minidsnetExt(R.CFG(aspp=0), labels=2, patch_type=patch), 31).cuda().eval()
    outs = m(a.cuda().bfloat16(), b.cuda().bfloat16())
    loss = TN.train_loss(outs, seg.cuda(), disp.cuda()); loss.backward()
    print(tag, "loss", float(loss), float(gold[tag + ".eval.loss"]))
    for k, v in sorted(gn(m).items()):
        key = "%s.eval.gnorm.%s" % (tag, k)
        if key in gold.files:
            print("   %-20s %.4g vs %.4g  ratio %.3f" % (k, v, float(gold[key]), v / max(1e-30, float(gold[key]))))
gold = np.load(os.path.join(G, "dsnet.npz"))
m = fill_state_dict(N.dsnet(R.CFG(), labels=2), 61).cuda().eval()
x, y = rand_input(61, "left", (2, 3, 256, 256)).cuda(), rand_input(61, "right", (2, 3, 256, 256)).cuda()
sg = F.one_hot((rand_input(61, "seg", (2, 256, 256)) > 0.5).long(), 2).permute(0, 3, 1, 2).float().cuda()
dp = rand_input(61, "disp", (2, 1, 256, 256), 0.0, 8.0).cuda()
outs = m(x.bfloat16(), y.bfloat16())
loss = torch.mean(torch.sum(-sg * outs[0].float(), 1)) + torch.mean(torch.sum(-sg * outs[2].float(), 1)) + F.l1_loss(outs[1].float(), dp) + F.l1_loss(outs[3].float(), dp)
loss.backward()
print("dsnet loss", float(loss), float(gold["dsnet.eval.loss"]))
for k, v in sorted(gn(m).items()):
    key = "dsnet.eval.gnorm.%s" % k
    if key in gold.files:
        print("   %-20s %.4g vs %.4g  ratio %.3f" % (k, v, float(gold[key]), v / max(1e-30, float(gold[key]))))
gold = np.load(os.path.join(G, "psmnet.npz"))
m = TPS._load_eval_stats(fill_state_dict(PSMNet(64), 41), gold).cuda().eval()
x, y = rand_input(41, "left", (2, 3, 256, 256)).cuda(), rand_input(41, "right", (2, 3, 256, 256)).cuda()
dp = rand_input(41, "disp", (2, 256, 256), 0.0, 40.0).cuda()
o = m(x.bfloat16(), y.bfloat16()); o = o if isinstance(o, tuple) else (o,)
loss = sum(F.l1_loss(t.float(), dp) for t in o) / len(o); loss.backward()
print("psm64 loss", float(loss), float(gold["psm64.eval.loss"]))
for k, v in sorted(gn(m).items()):
    key = "psm64.eval.gnorm.%s" % k
    if key in gold.files:
        print("   %-20s %.4g vs %.4g  ratio %.3f" % (k, v, float(gold[key]), v / max(1e-30, float(gold[key]))))
