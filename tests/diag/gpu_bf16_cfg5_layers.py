"""Where does the bf16 eval-mode error of the 19-class network come from?  Per-module rel L2 (bf16 vs f32 run, same weights)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N
import test_nets as TN, test_parity_r2 as TP
gold = np.load(os.path.join(TN.GDIR, "cfg5.npz"))
tag = sys.argv[1] if len(sys.argv) > 1 else "a0_hanet_l19"
recs = {}
def run(dt):
    m, x, y, pos, seg5, disp5 = TP._cfg5_case(gold, tag, N.minidsnetExt, "cuda")
    out = {}
    hs = []
    for name, mod in m.named_modules():
        if name.count(".") <= (1 if name.startswith("segNet") or name.startswith("hanet") else 0) and name:
            def hook(mod, inp, o, name=name):
                t = o[0] if isinstance(o, (tuple, list)) else o
                if torch.is_tensor(t):
                    out.setdefault(name, []).append(t.detach().float().clone())
            hs.append(mod.register_forward_hook(hook))
    with torch.no_grad():
        m(x.to(dt), y.to(dt), pos)
    return out
a = run(torch.float32); b = run(torch.bfloat16)
for k in a:
    for i, (u, v) in enumerate(zip(a[k], b.get(k, []))):
        if u.shape == v.shape:
            print("%-40s #%d %-24s rel %.4f  |f32| %.4g absmax %.4g" % (k, i, tuple(u.shape), float((u - v).norm() / u.norm().clamp_min(1e-20)), float(u.norm() / u.numel() ** 0.5), float(u.abs().max())))
