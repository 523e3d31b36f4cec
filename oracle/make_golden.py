"""ORACLE fixture generator (test infrastructure; runs ONLY in the build container).

Imports the reference's own PyTorch modules from /root/reference (read-only, no
bytecode written), fills them with the deterministic weights of
oracle/detweights.py, runs them on deterministic inputs on the CPU and stores
small .npz fixtures (inputs are regenerated from seeds; outputs are strided
samples + summary statistics) under tests/golden/.  The reference itself never
travels: tests on the GPU box only read the .npz files.

Stubs: `torchvision`, `efficientnet_pytorch`, `cv2` are imported by the reference
but unused on this path; `spatial_correlation_sampler` is an absent third-party
extension, replaced here by the oracle's restatement (oracle/ref_models.py,
oracle/corr_ref.c) — fixtures that pass through it are labelled
corr="assumed-semantics".

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = os.environ.get("SDHIP_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import ref_models as R  # noqa: E402
from oracle.detweights import fill_state_dict, rand_input, randn_input  # noqa: E402


def _install_stubs():
    for name in ("torchvision", "torchvision.models", "torchvision.transforms", "torchvision.datasets", "cv2"):
        sys.modules.setdefault(name, types.ModuleType(name))
    for sub in ("models", "transforms", "datasets"):
        setattr(sys.modules["torchvision"], sub, sys.modules["torchvision." + sub])
    sys.modules["cv2"].imwrite = lambda *a, **k: True   # GetSegMetricsNp / GetDispMetricsNp dump JPEGs; not part of the results
    tvf = types.ModuleType("torchvision.transforms.functional")   # util/utilTorchGate.py:38 imports `pad` (edge losses: not on this path)
    tvf.pad = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("torchvision stub"))
    sys.modules["torchvision.transforms"].__path__ = []
    sys.modules.setdefault("torchvision.transforms.functional", tvf)
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    eff = types.ModuleType("efficientnet_pytorch")
    eff.EfficientNet = type("EfficientNet", (), {})
    sys.modules["efficientnet_pytorch"] = eff
    scs = types.ModuleType("spatial_correlation_sampler")
    scs.SpatialCorrelationSampler = R.SpatialCorrelationSampler
    sys.modules["spatial_correlation_sampler"] = scs
    torch.Tensor.cuda = lambda self, *a, **k: self  # PSMNet / HANet hard-code .cuda()
    sys.path.insert(0, REF)


def sample(t, stride=8):
    """Strided sample + summary statistics of a tensor (what the fixtures store)."""
    t = t.detach().float()
    idx = tuple(slice(None, None, stride if (d >= t.dim() - 2 and t.shape[d] > 16) else
                      (4 if (d == 1 and t.dim() == 4 and t.shape[1] >= 64) else 1)) for d in range(t.dim()))
    return {"sample": t[idx].contiguous().numpy(), "mean": np.float64(t.double().mean()),
            "absmean": np.float64(t.double().abs().mean()), "l2": np.float64(t.double().pow(2).sum().sqrt())}


def flat(prefix, d):
    return {"%s.%s" % (prefix, k): v for k, v in d.items()}


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


def train_loss(outs, seg, disp):
    """The loss of the timed step (torch_implementation.py:279,293,304,325) without Lovasz: CE(seg1)+CE(seg2)+L1(disp)."""
    seg1, d1, seg2, _ = outs
    ce = lambda y: torch.mean(torch.sum(-seg * F.log_softmax(y, 1), 1))  # util/utilTorchLoss.py:373-378
    return ce(seg1) + ce(seg2) + F.l1_loss(d1, disp)


def grad_norms(model, depth=1):
    acc = {}
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        key = ".".join(k.split(".")[:depth])
        acc[key] = acc.get(key, 0.0) + float(p.grad.double().pow(2).sum())
    return {k: np.float64(np.sqrt(v)) for k, v in acc.items()}


# --------------------------------------------------------------------------- per-op fixtures
def gen_ops():
    from models import torch_model as TM
    from models import dsnet_t2 as D
    arrays = {}
    cases = [  # (name, cin, cout, k, stride, dil, H, W)
        ("c1", 8, 16, 1, 1, 1, 9, 11), ("c3", 16, 8, 3, 1, 1, 12, 17), ("c5", 8, 8, 5, 1, 1, 13, 16),
        ("c7s2", 3, 8, 7, 2, 1, 17, 20), ("c5d2", 3, 1, 5, 1, 2, 16, 19), ("c3s2", 8, 8, 3, 2, 1, 15, 16)]
    for name, ci, co, k, s, d, H, W in cases:
        ref = fill_state_dict(TM.conv2dSame(ci, co, k, s, 'same', d, bias=True), 11)
        mine = R.conv2dSame(ci, co, k, s, 'same', d, bias=True)
        mine.load_state_dict(ref.state_dict())
        x = randn_input(11, name, (2, ci, H, W)).requires_grad_(True)
        y = ref(x)
        g = randn_input(12, name, tuple(y.shape))
        y.backward(g)
        arrays.update({"conv.%s.y" % name: y.detach().numpy(), "conv.%s.gx" % name: x.grad.numpy(),
                       "conv.%s.gw" % name: ref.c2d.weight.grad.numpy(), "conv.%s.gb" % name: ref.c2d.bias.grad.numpy()})
        assert torch.allclose(mine(x), y, atol=1e-6)
    dcases = [("d3", 8, 16, 3, 1, 12, 17), ("d5", 8, 4, 5, 1, 13, 16), ("d3s2", 8, 8, 3, 2, 7, 9), ("d5s2", 4, 8, 5, 2, 6, 8)]
    for name, ci, co, k, s, H, W in dcases:
        ref = fill_state_dict(TM.ConvTranspose2dSame(ci, co, k, s, 'same', 1, bias=True), 13)
        mine = R.ConvTranspose2dSame(ci, co, k, s, 'same', 1, bias=True)
        mine.load_state_dict(ref.state_dict())
        x = randn_input(13, name, (2, ci, H, W)).requires_grad_(True)
        y = ref(x)
        g = randn_input(14, name, tuple(y.shape))
        y.backward(g)
        arrays.update({"deconv.%s.y" % name: y.detach().numpy(), "deconv.%s.gx" % name: x.grad.numpy(),
                       "deconv.%s.gw" % name: ref.ct2d.weight.grad.numpy(), "deconv.%s.gb" % name: ref.ct2d.bias.grad.numpy()})
        assert torch.allclose(mine(x), y, atol=1e-6)
    # convbn / deconvbn / Conv2DownUp in train mode (batch statistics + running stats)
    for name, ctor, args, shape in [
            ("convbn", D.convbn, (8, 16, 3, 1, 'same', 1), (4, 8, 10, 12)),
            ("deconvbn", D.deconvbn, (8, 8, 5, 1, 'same', 1), (4, 8, 10, 12)),
            ("cdu_last", lambda: D.Conv2DownUp(8, 16, 3, True), (), (2, 8, 12, 16)),
            ("cdu_nolast", lambda: D.Conv2DownUp(16, 8, 5, False), (), (2, 16, 12, 16))]:
        ref = fill_state_dict(ctor(*args), 15).train()
        x = randn_input(15, name, shape).requires_grad_(True)
        y = ref(x)
        g = randn_input(16, name, tuple(y.shape))
        y.backward(g)
        arrays.update({"%s.y" % name: y.detach().numpy(), "%s.gx" % name: x.grad.numpy()})
        for k, v in ref.state_dict().items():
            if k.endswith("running_mean") or k.endswith("running_var"):
                arrays["%s.state.%s" % (name, k)] = v.numpy().copy()
        for k, p in ref.named_parameters():
            if p.grad is not None:  # Conv2DownUp(lastLayer=False) never runs d5
                arrays["%s.grad.%s" % (name, k)] = p.grad.numpy().copy()
    save("ops", **arrays)


def gen_backbone():
    from models import densenet as DN
    from models import dsnet_t2 as D
    from models import aspp as A
    arrays = {}
    ref = fill_state_dict(DN.densenet121(pretrained=False), 21).train()
    x = rand_input(21, "img", (2, 3, 256, 256))
    taps = ref(x)
    for i, t in enumerate(taps):
        arrays.update(flat("densenet.tap%d" % i, sample(t, 8)))
    arrays["densenet.norm5.running_mean"] = ref.norm5.running_mean.numpy().copy()
    mine = R.densenet121()
    mine.load_state_dict(ref.state_dict())
    mine.train()
    for a, b in zip(mine(x), taps):
        assert torch.allclose(a, b, atol=1e-4, rtol=1e-4), float((a - b).abs().max())

    ref = fill_state_dict(D.piramidNet2(False, 'densenet'), 22).train()
    x = rand_input(22, "img", (2, 3, 256, 512))
    outs = ref(x)
    for i, t in enumerate(outs):
        arrays.update(flat("pyramid2.out%d" % i, sample(t, 8)))
    for tag, ctor in (("aspp_a1", lambda: A.build_aspp('densenet_a1', 32)), ("aspp_a3", lambda: A.build_aspp('densenet_a3', 32))):
        ref = fill_state_dict(ctor(), 23).eval()   # eval: Dropout(0.5) off, running statistics
        cin = 128 if tag == "aspp_a1" else 512
        x = randn_input(23, tag, (2, cin, 16, 24)).requires_grad_(True)
        y = ref(x)
        y.backward(randn_input(24, tag, tuple(y.shape)))
        arrays.update(flat(tag + ".y", sample(y, 2)))
        arrays.update(flat(tag + ".gx", sample(x.grad, 2)))
    save("backbone", **arrays)


def gen_nets():
    from models import dsnet_t2 as D
    arrays = {}
    cfgs = [("mini_a0", dict(aspp=0), '1dcorr'), ("mini_a1", dict(aspp=1), '1dcorr'), ("mini_a2", dict(aspp=2), '1dcorr'),
            ("mini_a0_2d", dict(aspp=0), '')]
    for tag, kw, patch in cfgs:
        cfg = R.CFG(**kw)
        for mode in ("train", "eval"):
            if mode == "train" and kw.get("aspp"):
                continue  # ASPP Dropout(0.5) is stochastic in train mode (models/aspp.py:79,95)
            ref = fill_state_dict(D.minidsnetExt(cfg, labels=2, pretrained=False, patch_type=patch, backbone='densenet'), 31)
            ref.train() if mode == "train" else ref.eval()
            a, b = rand_input(31, "left", (2, 3, 256, 256)), rand_input(31, "right", (2, 3, 256, 256))
            seg = F.one_hot((rand_input(31, "seg", (2, 256, 256)) > 0.5).long(), 2).permute(0, 3, 1, 2).float()
            disp = rand_input(31, "disp", (2, 1, 256, 256), 0.0, 8.0)
            outs = ref(a, b)
            loss = train_loss(outs, seg, disp)
            loss.backward()
            p = "%s.%s" % (tag, mode)
            for i, name in enumerate(("seg1", "disp", "seg2")):
                arrays.update(flat("%s.%s" % (p, name), sample(outs[i], 8)))
            arrays["%s.loss" % p] = np.float64(loss.item())
            for k, v in grad_norms(ref).items():
                arrays["%s.gnorm.%s" % (p, k)] = v
            if mode == "train":
                arrays["%s.rm.norm5" % p] = ref.resnet_features.resnet_features.norm5.running_mean.numpy().copy()
                arrays["%s.rv.norm5" % p] = ref.resnet_features.resnet_features.norm5.running_var.numpy().copy()
            # the oracle restatement must reproduce the reference here and now
            mine = R.minidsnetExt(cfg, labels=2, patch_type=patch)
            mine.load_state_dict(ref.state_dict() if mode == "eval" else fill_state_dict(
                D.minidsnetExt(cfg, labels=2, pretrained=False, patch_type=patch, backbone='densenet'), 31).state_dict())
            mine.train() if mode == "train" else mine.eval()
            mo = mine(a, b)
            for x, y in zip(mo, outs):
                err = float((x - y).abs().max())
                assert err < 2e-4, (tag, mode, err)
            print(tag, mode, "oracle==reference, loss", loss.item())
    arrays["meta.corr"] = np.array("assumed-semantics")
    save("nets", **arrays)


def gen_minidsnet():
    """`sdnet_mini` (models/dsnet_t2.py:825-913, util/utilLoadNetwork.py:10) train + eval, and the edge-channel variants
    (`-edges 1`: 4-channel inputs, include_edges=True) of minidsnet (2-D correlation) and minidsnetExt, eval."""
    from models import dsnet_t2 as D
    arrays = {}
    cases = [("mini", "minidsnet", '1dcorr', False, ("train", "eval")), ("mini_edges_2d", "minidsnet", '', True, ("eval",)),
             ("ext_edges", "minidsnetExt", '1dcorr', True, ("eval",))]
    for tag, cls, patch, edges, modes in cases:
        def make(mod):
            if cls == "minidsnet":
                return mod.minidsnet(R.CFG(), labels=2, pretrained=False, patch_type=patch, include_edges=edges)
            kw = dict(backbone='densenet') if mod is D else {}
            return mod.minidsnetExt(R.CFG(aspp=0), labels=2, pretrained=False, patch_type=patch, include_edges=edges, **kw)
        for mode in modes:
            ref = fill_state_dict(make(D), 33)
            ref.train() if mode == "train" else ref.eval()
            nc = 4 if edges else 3
            a, b = rand_input(33, "left", (2, nc, 256, 256)), rand_input(33, "right", (2, nc, 256, 256))
            seg = F.one_hot((rand_input(33, "seg", (2, 256, 256)) > 0.5).long(), 2).permute(0, 3, 1, 2).float()
            disp = rand_input(33, "disp", (2, 1, 256, 256), 0.0, 8.0)
            outs = ref(a, b)
            loss = train_loss(outs, seg, disp)
            loss.backward()
            p = "%s.%s" % (tag, mode)
            names = ("seg1", "disp") if cls == "minidsnet" else ("seg1", "disp", "seg2")
            for i, name in enumerate(names):
                arrays.update(flat("%s.%s" % (p, name), sample(outs[i], 8)))
            arrays["%s.loss" % p] = np.float64(loss.item())
            for k, v in grad_norms(ref).items():
                arrays["%s.gnorm.%s" % (p, k)] = v
            if mode == "train":
                bn = ref.conv2d_ba3[0].layers[1]      # an auxiliary branch nothing consumes: its statistics still move
                arrays["%s.rm.ba3" % p] = bn.running_mean.numpy().copy()
                arrays["%s.rv.ba3" % p] = bn.running_var.numpy().copy()
            mine = make(R)
            mine.load_state_dict(fill_state_dict(make(D), 33).state_dict())
            mine.train() if mode == "train" else mine.eval()
            for x, y in zip(mine(a, b), outs):
                err = float((x - y).abs().max())
                assert err < 2e-4, (tag, mode, err)
            print(tag, mode, "oracle==reference, loss", loss.item())
    arrays["meta.corr"] = np.array("assumed-semantics")
    save("minidsnet", **arrays)


def _pos(B, H, W):
    """The position maps of util/torch_implementation.py:139-143 at a reduced size (row index // 8 semantics kept:
    values 0..127 over the image height)."""
    h = (torch.arange(0, H) * 1024 // H).unsqueeze(0).unsqueeze(2).expand(B, -1, W) // 8
    w = (torch.arange(0, W) * 2048 // W).unsqueeze(0).unsqueeze(1).expand(B, H, -1) // 16
    return h, w


def gen_hanet():
    """HANet head alone (train: batch-statistics BatchNorm1d, dropout off; eval) and inside minidsnetExt (eval)."""
    torch.Tensor.cuda = lambda self, *a, **k: self      # PosEncoding1D hard-codes .cuda() (models_hanet/PosEmbedding.py:54,68)
    from models_hanet.HANet import HANet_Conv
    from models import dsnet_t2 as D
    arrays = {}
    for mode in ("train", "eval"):
        ref = fill_state_dict(HANet_Conv(64, 5, pooling='max', pos_rfactor=2, dropout_prob=0.0 if mode == "train" else 0.1), 41)
        ref.train() if mode == "train" else ref.eval()
        x = randn_input(41, "hx", (2, 64, 128, 24)).requires_grad_(True)
        out = randn_input(41, "hout", (2, 5, 96, 40)).requires_grad_(True)
        pos = _pos(2, 256, 8)
        y, logits = ref(x, out, pos, attention_loss=True)
        gy = randn_input(42, "hgy", tuple(y.shape))
        (y * gy).sum().backward()
        p = "hanet.%s" % mode
        arrays.update(flat(p + ".y", sample(y, 4)))
        arrays[p + ".logits"] = logits.detach().numpy().copy()
        arrays[p + ".gx.norm"] = np.float64(x.grad.norm().item())
        arrays.update(flat(p + ".gx", sample(x.grad, 4)))
        arrays.update(flat(p + ".gout", sample(out.grad, 4)))
        for k, v in ref.named_parameters():
            if v.grad is not None:
                arrays["%s.gw.%s" % (p, k)] = v.grad.numpy().copy()
        if mode == "train":
            arrays[p + ".rm2"] = ref.attention_second[1].running_mean.numpy().copy()
            arrays[p + ".rv2"] = ref.attention_second[1].running_var.numpy().copy()
        mine = R.HANet_Conv(64, 5, pooling='max', pos_rfactor=2, dropout_prob=0.0 if mode == "train" else 0.1)
        mine.load_state_dict(fill_state_dict(HANet_Conv(64, 5, pooling='max', pos_rfactor=2, dropout_prob=0.0), 41).state_dict())
        mine.train() if mode == "train" else mine.eval()
        my, ml = mine(x.detach(), out.detach(), pos, attention_loss=True)
        assert float((my - y).abs().max()) < 1e-5 and float((ml - logits).abs().max()) < 1e-5, mode
        print("hanet", mode, "oracle==reference")
    # inside the network (eval: Dropout2d(0.1) inactive)
    cfg = R.CFG(aspp=0, hanet=1)
    ref = fill_state_dict(D.minidsnetExt(cfg, labels=2, pretrained=False, patch_type='1dcorr', backbone='densenet'), 43).eval()
    a, b = rand_input(43, "left", (2, 3, 256, 256)), rand_input(43, "right", (2, 3, 256, 256))
    pos = _pos(2, 256, 256)
    with torch.no_grad():
        outs = ref(a, b, pos)
    for i, name in enumerate(("seg1", "disp", "seg2")):
        arrays.update(flat("mini_hanet.eval.%s" % name, sample(outs[i], 8)))
    mine = R.minidsnetExt(cfg, labels=2, patch_type='1dcorr')
    mine.load_state_dict(ref.state_dict())
    mine.eval()
    with torch.no_grad():
        mo = mine(a, b, pos)
    for x_, y_ in zip(mo, outs):
        assert float((x_ - y_).abs().max()) < 2e-4
    print("mini_hanet eval oracle==reference")
    arrays["meta.corr"] = np.array("assumed-semantics")
    save("hanet", **arrays)


def gen_metrics():
    """The host-side step metrics of lossSeg_fn / lossDisp_fn (losses/multiLosses.py:116-125,146-154), produced by the
    reference's own numpy / sklearn functions.  Inputs are stored (they are small)."""
    import warnings
    from util import utilTorchLoss as UL
    rng = np.random.default_rng(7)
    arrays = {}
    cases = [("roses", 2, 2, 2, 24, 40, 1.0, False), ("city", 19, 20, 2, 16, 24, 192.0, True), ("one", 2, 2, 1, 8, 8, 1.0, False)]
    for name, L, Ct, B, H, W, max_disp, mask_invalid in cases:
        logits = rng.normal(0, 2, (B, L, H, W)).astype(np.float32)
        logits[0, 1, 0, :4] = [0.0, 1.0, -1.0, 1.0]                        # exact 0 / 1 logits exercise the threshold and branch-mask rules
        cls = rng.integers(0, Ct, (B, H, W))
        seg_full = np.eye(Ct, dtype=np.float32)[cls].transpose(0, 3, 1, 2).copy()
        if name == "roses_emptyrow":
            seg_full[:, :, ::3, 1::4] = 0.0
        disp = (rng.uniform(0, 8, (B, 1, H, W)) * (rng.uniform(0, 1, (B, 1, H, W)) > (0.3 if mask_invalid else -1))).astype(np.float32)
        if not mask_invalid:
            disp += np.float32(0.1)
        disp_pred = (disp + rng.normal(0, 1.5 / max_disp ** 0.5, disp.shape)).astype(np.float32)
        logsm = F.log_softmax(torch.from_numpy(logits), 1).numpy()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            acc, conf, _ = UL.SegAccuracyNp(logsm, seg_full, L)
            prec, rec, f1, bf1 = UL.GetSegMetricsNp(logits.copy(), seg_full.copy(), L, num_image=0)
            zeros = (disp > 0) * np.float32(1.0) if mask_invalid else 1.0     # multiLosses.py:134-139,146-148
            dp, dg = disp_pred * zeros, disp * zeros
            err, val = UL.unnormalizedErrorNP(dp, dg, max_disp)
            rmse, sqrel, brmse, bsqrel = UL.GetDispMetricsNp(dp.copy(), dg.copy(), seg_full, num_image=0)
        arrays.update(flat(name, dict(logits=logits, seg_full=seg_full, disp=disp, disp_pred=disp_pred, labels=np.int64(L),
                                      max_disp=np.float64(max_disp), mask_invalid=np.int64(mask_invalid),
                                      pixelAcc=np.float64(acc), conf_matrix=conf.astype(np.int64), pixelPrec=np.float64(prec),
                                      pixelRecall=np.float64(rec), pixelF1=np.float64(f1), pixelBF1=np.float64(bf1),
                                      err=np.float64(err), val_pxl=np.float64(val), dispRMSE=np.float64(rmse),
                                      dispSqRel=np.float64(sqrel), BdispRMSE=np.float64(brmse), BdispSqRel=np.float64(bsqrel))))
    save("metrics", **arrays)


def gen_keys():
    """Ordered state_dict keys (+shapes) and ordered parameter names of the reference's networks: the checkpoint surface
    (`state_dict` keys, and torch.optim.Adam state indexed by parameter order — torch_implementation.py:915-934)."""
    import importlib
    import json
    from models import dsnet_t2 as D
    SH = importlib.import_module("models_psmnet.stackhourglass")
    nets = {
        "mini_a0": lambda: D.minidsnetExt(R.CFG(aspp=0), labels=2, pretrained=False, patch_type='1dcorr', backbone='densenet'),
        "mini_a1": lambda: D.minidsnetExt(R.CFG(aspp=1), labels=2, pretrained=False, patch_type='1dcorr', backbone='densenet'),
        "mini_a2_hanet": lambda: D.minidsnetExt(R.CFG(aspp=2, hanet=1), labels=19, pretrained=False, patch_type='1dcorr', backbone='densenet'),
        "dsnet": lambda: D.dsnet(R.CFG(), labels=2, pretrained=False),
        "psmnet192": lambda: SH.PSMNet(192),
        "minidsnet": lambda: D.minidsnet(R.CFG(), labels=2, pretrained=False, patch_type='1dcorr'),
        "mini_a0_edges": lambda: D.minidsnetExt(R.CFG(aspp=0), labels=2, pretrained=False, patch_type='1dcorr', include_edges=True,
                                                backbone='densenet'),
    }
    out = {}
    for name, ctor in nets.items():
        m = ctor()
        out[name] = {"state_dict": [[k, list(v.shape)] for k, v in m.state_dict().items()],
                     "parameters": [k for k, _ in m.named_parameters()]}
        print(name, len(out[name]["state_dict"]), "keys,", len(out[name]["parameters"]), "parameters")
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "keys.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))


def _install_loader_stubs():
    """util/utilTorchDataLoader.py imports skimage / cv2 / torchvision at module level; none is installed here.  The
    fixture only covers what happens AFTER the containers are decoded, so the two decoders it calls are routed to PIL
    (`skimage.io.imread`, `cv2.imread(path, -1)`), and the Sobel edge map (an output this build does not produce,
    `-edges 0`) is a zero map.  Nothing else from these packages runs on the path."""
    from PIL import Image
    sk = types.ModuleType("skimage")
    for sub in ("io", "transform", "filters", "measure", "color", "morphology"):
        m = types.ModuleType("skimage." + sub)
        sys.modules["skimage." + sub] = m
        setattr(sk, sub, m)
    sys.modules["skimage"] = sk
    sk.io.imread = lambda path: np.asarray(Image.open(path))
    sk.filters.sobel = lambda img: np.zeros(np.asarray(img).shape[:2], dtype=np.float64)
    sk.measure.label = sk.color.rgb2gray = sk.morphology.dilation = sk.morphology.square = lambda *a, **k: a[0]
    sys.modules["cv2"].imread = lambda path, flag=-1: np.asarray(Image.open(path))
    tvf = types.ModuleType("torchvision.transforms.functional")
    sys.modules["torchvision.transforms.functional"] = tvf
    sys.modules["torchvision.transforms"].functional = tvf
    sys.modules["torchvision.transforms"].Compose = lambda ts: (lambda s: [s := t(s) for t in ts][-1])


def _write_pfm(path, img, little=True):
    """Grey PFM as the format defines it (rows bottom-up); both byte orders, which readPFM accepts (utilIOPfm.py:89-93)."""
    h, w = img.shape
    with open(path, "wb") as f:
        f.write(b"Pf\n%d %d\n%s\n" % (w, h, b"-1.000000" if little else b"1.000000"))
        f.write(np.flipud(img).astype("<f4" if little else ">f4").tobytes())


def gen_data():
    """Samples produced by the reference's own CustomDataset.__getitem__ / RandomCrop / ToTensor / readPFM
    (util/utilTorchDataLoader.py, util/utilIOPfm.py) from synthetic files; inputs are stored raw."""
    import tempfile
    from PIL import Image
    _install_loader_stubs()
    from util import utilTorchDataLoader as DL
    from util.utilCityscape import id2label
    rng = np.random.default_rng(11)
    arrays = {}
    H, W = 24, 40
    tmp = tempfile.mkdtemp(prefix="sdhip_golden_")
    cases = [("roses_linear", "roses", 2, 192, "linear", (0, 0), True, ((0., 0., 0.), (1., 1., 1.))),
             ("roses_crop", "roses", 2, 192, "linear", (16, 24), True, ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))),
             ("roses_sigmoid_be", "roses", 2, 12, "sigmoid", (0, 0), False, ((0., 0., 0.), (1., 1., 1.))),
             ("garden_tanh", "garden", 9, 12, "tanh", (16, 32), True, ((0., 0., 0.), (1., 1., 1.))),
             ("city_linear", "cityscapes", 19, 192, "linear", (16, 24), True, ((0., 0., 0.), (1., 1., 1.))),
             ("city_flip", "cityscapes", 19, 192, "linear", (16, 32), True, ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))),
             ("city_flip_whole", "cityscapes", 19, 192, "linear", (24, 40), True, ((0., 0., 0.), (1., 1., 1.))),
             # appended (the generator's draws for the cases above are unchanged): the numpy-only augmentations of RandomCrop
             ("roses_slice", "roses", 2, 192, "linear", (16, 24), True, ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))),
             ("garden_double", "garden", 9, 12, "tanh", (16, 32), True, ((0., 0., 0.), (1., 1., 1.))),
             ("city_slice_double_down", "cityscapes", 19, 192, "linear", (16, 24), True, ((0., 0., 0.), (1., 1., 1.))),
             ("kitti_band", "kitti", 19, 192, "linear", (8, 24), True, ((0., 0., 0.), (1., 1., 1.)))]
    OPTS = {"roses_slice": dict(slice=True), "garden_double": dict(double=True),
            "city_slice_double_down": dict(slice=True, double=True, down=True), "kitti_band": dict(band=True)}
    if not hasattr(np, "int"):
        np.int = int          # RandomCrop's flip uses the alias numpy removed in 1.24 (util/utilTorchDataLoader.py:485)
    for name, ds, n_labels, max_d, act, crop, little, norm in cases:
        flip = name.startswith("city_flip")
        opt = OPTS.get(name, {})
        left = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        right = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        if ds == "roses":
            seg = rng.choice(np.array([0, 127, 128, 129, 255], dtype=np.uint8), (H, W, 3))
        elif ds == "garden":
            seg = rng.integers(0, n_labels + 2, (H, W), dtype=np.uint8)          # 0 and n_labels+1 are in no class
        else:
            seg = rng.integers(0, 34, (H, W), dtype=np.uint8)
        inst = rng.integers(0, 4, (H, W), dtype=np.uint8)
        p = lambda f: os.path.join(tmp, name + "_" + f)
        Image.fromarray(left).save(p("l.png")); Image.fromarray(right).save(p("r.png"))
        Image.fromarray(seg).save(p("s.png")); Image.fromarray(inst).save(p("i.png"))
        if ds in ("roses", "garden"):
            depth = rng.uniform(0.5, 40, (H, W)).astype(np.float32)
            depth[rng.uniform(size=(H, W)) < 0.1] = 0.0
            depth[rng.uniform(size=(H, W)) < 0.05] = -1.0
            depth[0, 0], depth[0, 1] = np.nan, np.inf
            _write_pfm(p("d.pfm"), depth, little)
            dpath = p("d.pfm")
            arrays[name + ".depth_file"] = np.frombuffer(open(dpath, "rb").read(), dtype=np.uint8)
        else:
            d16 = rng.integers(0, 65536 if not flip else 256 * 9, (H, W), dtype=np.uint16)    # flip cases: disparities < 9 px
            d16[rng.uniform(size=(H, W)) < 0.2] = 0
            Image.fromarray(d16).save(p("d.png"))
            dpath = p("d.png")
            arrays[name + ".depth_u16"] = d16
        normalize = np.array(norm, dtype=np.float32)
        tf = DL.RandomCrop(list(crop), datasetName=ds, is_down=bool(opt.get("down")), sliceandSwitch=bool(opt.get("slice")),
                           augment_DoubleLeftImg=bool(opt.get("double")), focusPerson=False, resizeImg=False, flipHorizontal=flip)
        dset = DL.CustomDataset([(p("l.png"), p("r.png"))], [(dpath, p("s.png"), p("i.png"))], n_labels, max_d, ds, normalize,
                                output_activation=act, transform=tf, to_tensor=DL.ToTensor())
        seed = 123
        from pmt_learning_for_semantic_segmentation_and_disparity_amd import data as PD    # host-side draw helpers only (no GPU call)
        if flip:      # a seed whose draws (crop offsets, then the 50 % flip decision) end in "flip"
            for seed in range(123, 200):
                torch.manual_seed(seed)
                PD.draw_crop(H, W, list(crop), ds)
                if PD.draw_flip(ds, True):
                    break
        if opt.get("double") or opt.get("band"):
            # a seed whose draws end in the 10 % "double left" decision / in kitti's 80 % lower-band crop
            for seed in range(123, 400):
                torch.manual_seed(seed)
                top = PD.draw_crop(H, W, list(crop), ds, is_down=bool(opt.get("down")))[0]
                PD.draw_slice_and_switch(H, crop[0], bool(opt.get("slice")))
                if opt.get("double") and PD.draw_double_left(True):
                    break
                if opt.get("band") and top >= 1:
                    break
        torch.manual_seed(seed)
        s = dset[0]
        arrays.update(flat(name, dict(left_u8=left, right_u8=right, seg_u8=seg, n_labels=np.int64(n_labels), max_d=np.float64(max_d),
                                      crop=np.array(crop, dtype=np.int64), normalize=normalize, seed=np.int64(seed), flip=np.int64(flip),
                                      opt_slice=np.int64(bool(opt.get("slice"))), opt_double=np.int64(bool(opt.get("double"))),
                                      opt_down=np.int64(bool(opt.get("down"))),
                                      left=s["left"].numpy(), right=s["right"].numpy(), seg=s["seg"].numpy().astype(np.float32),
                                      disp=s["disp"].numpy().astype(np.float32))))
        arrays[name + ".dataset"] = np.array(ds)
        arrays[name + ".activation"] = np.array(act)
        print(name, {k: tuple(v.shape) for k, v in s.items() if hasattr(v, "shape")}, s["disp"].dtype, s["seg"].dtype)
    lut = np.full(256, 255, dtype=np.int64)
    for i, lab in id2label.items():
        if 0 <= i < 256:
            lut[i] = lab.trainId
    arrays["cityscapes.id2trainId"] = lut
    save("data", **arrays)
    import shutil
    shutil.rmtree(tmp)


def gen_dsnet():
    from models import dsnet_t2 as D
    arrays = {}
    for mode in ("train", "eval"):
        ref = fill_state_dict(D.dsnet(R.CFG(), labels=2, pretrained=False), 61)
        ref.train() if mode == "train" else ref.eval()
        a, b = rand_input(61, "left", (2, 3, 256, 256)), rand_input(61, "right", (2, 3, 256, 256))
        seg = F.one_hot((rand_input(61, "seg", (2, 256, 256)) > 0.5).long(), 2).permute(0, 3, 1, 2).float()
        disp = rand_input(61, "disp", (2, 1, 256, 256), 0.0, 8.0)
        outs = ref(a, b)
        # dsnet returns log-probabilities: NLL-style loss on them + L1 on both disparity heads
        loss = torch.mean(torch.sum(-seg * outs[0], 1)) + torch.mean(torch.sum(-seg * outs[2], 1)) + F.l1_loss(outs[1], disp) + F.l1_loss(outs[3], disp)
        loss.backward()
        p = "dsnet.%s" % mode
        for i, name in enumerate(("seg1", "disp", "seg2", "disp2")):
            arrays.update(flat("%s.%s" % (p, name), sample(outs[i], 8)))
        arrays["%s.loss" % p] = np.float64(loss.item())
        for k, v in grad_norms(ref).items():
            arrays["%s.gnorm.%s" % (p, k)] = v
        mine = R.dsnet(R.CFG(), labels=2)
        mine.load_state_dict(fill_state_dict(D.dsnet(R.CFG(), labels=2, pretrained=False), 61).state_dict() if mode == "train" else ref.state_dict())
        mine.train() if mode == "train" else mine.eval()
        for x, y in zip(mine(a, b), outs):
            err = float((x - y).abs().max())
            assert err < 2e-4 * max(1.0, float(y.abs().max())), (mode, err)
        print("dsnet", mode, "oracle==reference, loss", loss.item())
    arrays["meta.corr"] = np.array("assumed-semantics")
    save("dsnet", **arrays)


def gen_losses():
    """The reference's own loss functions (losses/multiLosses.py lossSeg_fn / lossDisp_fn, util/lovasz_losses.py) on small
    stored inputs: value and gradient w.r.t. the network outputs.  `cross_entropy lovasz_loss` is the shipped recipe
    (scripts/trainTorchImpl.sh:34); cityscapes adds the ignore channel and the disp > 0 mask."""
    import warnings
    from losses import multiLosses as ML
    from util import lovasz_losses as LV
    rng = np.random.default_rng(5)
    arrays = {}
    cases = [("roses", "roses", 2, 2, 2, 32, 32, None), ("garden_absent", "garden", 5, 5, 2, 16, 24, 3),
             ("city", "cityscapes", 19, 20, 2, 24, 32, 7), ("city_allvoid_class", "cityscapes", 19, 20, 1, 8, 8, None),
             # roses target with all-zero rows under `ignore=None`: argmax makes them class 0 and they count (appended last:
             # the generator's draws for the cases above are unchanged)
             ("roses_emptyrow", "roses", 2, 2, 2, 16, 16, None)]
    for name, ds, L, Ct, B, H, W, absent in cases:
        logits = rng.normal(0, 2, (B, L, H, W)).astype(np.float32)
        cls = rng.integers(0, Ct, (B, H, W))
        if absent is not None:
            cls[cls == absent] = (absent + 1) % L
        if name == "city_allvoid_class":
            cls[:] = 19                      # only void pixels: the Lovasz term and its gradient are zero
            cls[0, 0, :3] = [1, 1, 4]
        seg_full = np.eye(Ct, dtype=np.float32)[cls].transpose(0, 3, 1, 2).copy()
        if name == "roses_emptyrow":
            seg_full[:, :, ::3, 1::4] = 0.0
        disp = (rng.uniform(0, 8, (B, 1, H, W)) * (rng.uniform(0, 1, (B, 1, H, W)) > (0.3 if ds == "cityscapes" else -1))).astype(np.float32)
        disp_pred = (disp + rng.normal(0, 1.0, disp.shape)).astype(np.float32)
        cfg = types.SimpleNamespace(datasetName=ds, segWeight=0, outputType='segDisp')
        for tag, loss_type in (("ce_lovasz", ['cross_entropy', 'lovasz_loss']), ("ce", ['cross_entropy']), ("lovasz", ['lovasz_loss'])):
            # (copies: lossSeg_fn's metric block edits `seg_pred.detach().cpu().numpy()` in place, which on the CPU aliases the input)
            y = torch.from_numpy(logits.copy()).requires_grad_(True)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                out = ML.lossSeg_fn(loss_type, torch.from_numpy(seg_full.copy()), y, cfg, 0)
            loss = out[2]
            loss.backward()
            arrays["%s.%s.loss" % (name, tag)] = np.float64(loss.item())
            arrays["%s.%s.grad" % (name, tag)] = y.grad.numpy().copy()
        d = torch.from_numpy(disp_pred.copy()).requires_grad_(True)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = ML.lossDisp_fn([], None, torch.from_numpy(seg_full.copy()), torch.from_numpy(disp.copy()), d, 192.0, cfg, 0)
        out[2].backward()
        arrays["%s.l1.loss" % name] = np.float64(out[2].item())
        arrays["%s.l1.grad" % name] = d.grad.numpy().copy()
        # the Lovasz extension on its own (util/lovasz_losses.py:153-168), all three `classes` modes the function offers
        y = torch.from_numpy(logits.copy())
        lab = torch.from_numpy(seg_full.argmax(1))      # == cls, except where a one-hot row was cleared (roses_emptyrow)
        ign = None if ds in ("roses", "garden") else 19
        arrays["%s.lovasz_present" % name] = np.float64(float(LV.lovasz_softmax(F.softmax(y, 1), lab, ignore=ign)))
        arrays.update({"%s.logits" % name: logits, "%s.seg_full" % name: seg_full, "%s.disp" % name: disp,
                       "%s.disp_pred" % name: disp_pred, "%s.labels" % name: np.int64(L)})
        arrays["%s.dataset" % name] = np.array(ds)
        # the oracle restatement must agree here and now
        from oracle.losses_ref import train_loss_ref
        seg_t = torch.from_numpy(seg_full[:, :L])
        y1 = torch.from_numpy(logits).requires_grad_(True)
        dd = torch.from_numpy(disp_pred).requires_grad_(True)
        zero = torch.zeros_like(y1)
        mine = train_loss_ref(y1, dd, y1, seg_t, torch.from_numpy(disp), True, ds == "cityscapes", ds == "cityscapes")   # CE counted twice
        want = 2 * arrays["%s.ce.loss" % name] + arrays["%s.lovasz.loss" % name] + arrays["%s.l1.loss" % name]
        assert abs(float(mine) - want) < 1e-5 * max(1.0, abs(want)), (name, float(mine), want)
        mine.backward()
        gw = 2 * arrays["%s.ce.grad" % name] + arrays["%s.lovasz.grad" % name]
        assert float((y1.grad - torch.from_numpy(gw)).abs().max()) < 1e-6, name
        assert abs(arrays["%s.ce_lovasz.loss" % name] - arrays["%s.ce.loss" % name] - arrays["%s.lovasz.loss" % name]) < 1e-5, name
        assert float((dd.grad - torch.from_numpy(arrays["%s.l1.grad" % name])).abs().max()) < 1e-7, name
        print("losses", name, "oracle==reference", want)
    save("losses", **arrays)


def gen_dsnetnocorr():
    from models import dsnet_t2 as D
    arrays = {}
    for mode in ("train", "eval"):
        ref = fill_state_dict(D.dsnetnoCorr(R.CFG(), labels=2, pretrained=False), 71)
        ref.train() if mode == "train" else ref.eval()
        a, b = rand_input(71, "left", (2, 3, 256, 256)), rand_input(71, "right", (2, 3, 256, 256))
        seg = F.one_hot((rand_input(71, "seg", (2, 256, 256)) > 0.5).long(), 2).permute(0, 3, 1, 2).float()
        disp = rand_input(71, "disp", (2, 1, 256, 256), 0.0, 8.0)
        outs = ref(a, b)
        loss = torch.mean(torch.sum(-seg * outs[0], 1)) + torch.mean(torch.sum(-seg * outs[2], 1)) + F.l1_loss(outs[1], disp) + F.l1_loss(outs[3], disp)
        loss.backward()
        p = "dsnetnocorr.%s" % mode
        for i, name in enumerate(("seg1", "disp", "seg2", "disp2")):
            arrays.update(flat("%s.%s" % (p, name), sample(outs[i], 8)))
        arrays["%s.loss" % p] = np.float64(loss.item())
        for k, v in grad_norms(ref).items():
            arrays["%s.gnorm.%s" % (p, k)] = v
        mine = R.dsnetnoCorr(R.CFG(), labels=2)
        mine.load_state_dict(fill_state_dict(D.dsnetnoCorr(R.CFG(), labels=2, pretrained=False), 71).state_dict() if mode == "train" else ref.state_dict())
        mine.train() if mode == "train" else mine.eval()
        for x, y in zip(mine(a, b), outs):
            err = float((x - y).abs().max())
            assert err < 2e-4 * max(1.0, float(y.abs().max())), (mode, err)
        print("dsnetnoCorr", mode, "oracle==reference, loss", loss.item())
    save("dsnetnocorr", **arrays)


def gen_syncbn():
    """Statistics maths of synchronised BatchNorm on split batches: the reference's vendored `_compute_mean_std`
    (sync_batchnorm/batchnorm.py:114-126; dead code upstream but named by SURVEY 8c as the semantics statement) and the
    live path's torch BatchNorm on the joint batch (nn.SyncBatchNorm, torch_implementation.py:739)."""
    from sync_batchnorm.batchnorm import SynchronizedBatchNorm2d
    arrays = {}
    for name, B, C, H, W, parts in (("two_ranks", 4, 6, 5, 7, 2), ("four_ranks_tiny", 8, 3, 1, 2, 4)):
        x = randn_input(81, name, (B, C, H, W)) * 2.0 + 0.5
        bn = SynchronizedBatchNorm2d(C)
        bn.running_mean = randn_input(82, name + "rm", (C,))
        bn.running_var = rand_input(82, name + "rv", (C,), 0.5, 2.0)
        rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
        shards = x.chunk(parts, 0)
        sums = [(s.sum((0, 2, 3)), (s * s).sum((0, 2, 3)), s.numel() // C) for s in shards]      # what every replica sends (:62-64)
        sum_ = sum(t[0] for t in sums); ssum = sum(t[1] for t in sums); size = sum(t[2] for t in sums)
        mean, inv_std = bn._compute_mean_std(sum_, ssum, size)
        ref = torch.nn.BatchNorm2d(C)
        ref.running_mean.copy_(rm0); ref.running_var.copy_(rv0)
        y = ref.train()(x)
        arrays.update(flat(name, dict(x=x.numpy(), parts=np.int64(parts), rm0=rm0.numpy(), rv0=rv0.numpy(),
                                      mean=mean.numpy(), inv_std_clamp=inv_std.numpy(), rm_sync=bn.running_mean.numpy().copy(),
                                      rv_sync=bn.running_var.numpy().copy(), y_joint=y.detach().numpy(),
                                      rm_joint=ref.running_mean.numpy().copy(), rv_joint=ref.running_var.numpy().copy())))
        assert torch.allclose(bn.running_mean, ref.running_mean, atol=1e-6) and torch.allclose(bn.running_var, ref.running_var, atol=1e-5)
    save("syncbn", **arrays)


def gen_cfg5():
    """BASELINE config 5's network — minidsnetExt(aspp=2, hanet=1, labels=19) — and the (aspp=0, hanet=1, labels=19) variant
    in which the HANet head is actually applied (models/dsnet_t2.py:1287-1289), eval mode (ASPP Dropout(0.5) and HANet
    Dropout2d(0.1) are stochastic in train mode), with the cityscapes loss rule (void class, disp > 0 mask) and its
    gradients through the eval-mode network."""
    torch.Tensor.cuda = lambda self, *a, **k: self
    from models import dsnet_t2 as D
    from oracle.losses_ref import train_loss_ref
    arrays = {}
    for tag, kw in (("a2_hanet_l19", dict(aspp=2, hanet=1)), ("a0_hanet_l19", dict(aspp=0, hanet=1))):
        cfg = R.CFG(**kw)
        ref = fill_state_dict(D.minidsnetExt(cfg, labels=19, pretrained=False, patch_type='1dcorr', backbone='densenet'), 91)
        a, b = rand_input(91, "left", (2, 3, 256, 256)), rand_input(91, "right", (2, 3, 256, 256))
        pos = _pos(2, 256, 256)
        # running statistics as after training (batch statistics of this input; dropouts off while they are collected),
        # stored in the fixture: random running statistics blow the activations up through ~120 layers (loss ~ 1e4)
        bns = [m for m in ref.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
        drops = [m for m in ref.modules() if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d))]
        keep_p = [m.p for m in drops]
        for m in bns:
            m.reset_running_stats(); m.momentum = None
        for m in drops:
            m.p = 0.0
        ref.train()
        with torch.no_grad():
            ref(a, b, pos)
        for m in bns:
            m.momentum = 0.1
        for m, pp in zip(drops, keep_p):
            m.p = pp
        ref.eval()
        for k, v in ref.state_dict().items():
            if k.endswith("running_mean") or k.endswith("running_var"):
                arrays["%s.state.%s" % (tag, k)] = v.numpy().astype(np.float32).copy()
        cls = (rand_input(91, "cls", (2, 256, 256)) * 20).long().clamp(0, 19)           # 19 = void
        seg = F.one_hot(cls, 20).permute(0, 3, 1, 2).float()[:, :19].contiguous()
        disp = rand_input(91, "disp", (2, 1, 256, 256), 0.0, 8.0) * (rand_input(91, "dmask", (2, 1, 256, 256)) > 0.3).float()
        outs = ref(a, b, pos)
        loss = train_loss_ref(outs[0], outs[1], outs[2], seg, disp, True, True, True)
        loss.backward()
        for i, name in enumerate(("seg1", "disp", "seg2")):
            arrays.update(flat("%s.eval.%s" % (tag, name), sample(outs[i], 8)))
        arrays["%s.eval.loss" % tag] = np.float64(loss.item())
        for k, v in grad_norms(ref).items():
            arrays["%s.eval.gnorm.%s" % (tag, k)] = v
        mine = R.minidsnetExt(cfg, labels=19, patch_type='1dcorr')
        mine.load_state_dict(ref.state_dict())
        mine.eval()
        with torch.no_grad():
            mo = mine(a, b, pos)
        for x_, y_ in zip(mo, outs):
            assert float((x_ - y_).abs().max()) < 2e-4 * max(1.0, float(y_.abs().max())), tag
        print("cfg5", tag, "oracle==reference, loss", loss.item())
    arrays["meta.corr"] = np.array("assumed-semantics")
    save("cfg5", **arrays)


def gen_psmnet():
    import importlib
    SH = importlib.import_module("models_psmnet.stackhourglass")   # (the package re-exports the class under this name)
    arrays = {}
    for mode in ("train", "eval"):
        ref = fill_state_dict(SH.PSMNet(64), 41)
        a, b = rand_input(41, "left", (2, 3, 256, 256)), rand_input(41, "right", (2, 3, 256, 256))
        disp = rand_input(41, "disp", (2, 256, 256), 0.0, 40.0)
        if mode == "eval":
            # running statistics of a trained network, not random ones (random statistics saturate the softmax over
            # disparities and make the eval prediction a step function of 1e-6 cost differences): one train-mode forward
            # with cumulative-average momentum sets them to the batch statistics of this input; they are stored.
            bns = [m for m in ref.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
            for m in bns:
                m.reset_running_stats()
                m.momentum = None
            ref.train()
            with torch.no_grad():
                ref(a, b)
            for m in bns:
                m.momentum = 0.1
            for k, v in ref.state_dict().items():
                if k.endswith("running_mean") or k.endswith("running_var"):
                    arrays["psm64.eval.state.%s" % k] = v.numpy().copy()
        ref.train() if mode == "train" else ref.eval()
        outs = ref(a, b)
        outs = outs if isinstance(outs, tuple) else (outs,)
        loss = sum(F.l1_loss(o, disp) for o in outs) / len(outs)   # build-defined PSMNet loss: mean L1 of the predictions
        loss.backward()
        p = "psm64.%s" % mode
        for i, o in enumerate(outs):
            arrays.update(flat("%s.pred%d" % (p, i), sample(o, 8)))
        arrays["%s.loss" % p] = np.float64(loss.item())
        for k, v in grad_norms(ref).items():
            arrays["%s.gnorm.%s" % (p, k)] = v
        mine = R.PSMNet(64)
        mine.load_state_dict(fill_state_dict(SH.PSMNet(64), 41).state_dict() if mode == "train" else ref.state_dict())
        mine.train() if mode == "train" else mine.eval()
        mo = mine(a, b)
        mo = mo if isinstance(mo, tuple) else (mo,)
        for x, y in zip(mo, outs):
            err = float((x - y).abs().max())
            assert err < 1e-3, (mode, err)
        print("psmnet", mode, "oracle==reference, loss", loss.item())
    save("psmnet", **arrays)


if __name__ == "__main__":
    _install_stubs()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["ops", "backbone", "nets", "psmnet", "dsnet", "hanet", "metrics", "keys", "data", "losses", "dsnetnocorr", "syncbn", "cfg5",
                             "minidsnet"]
    if "minidsnet" in which:
        gen_minidsnet()
    if "losses" in which:
        gen_losses()
    if "syncbn" in which:
        gen_syncbn()
    if "cfg5" in which:
        gen_cfg5()
    if "dsnetnocorr" in which:
        gen_dsnetnocorr()
    if "ops" in which:
        gen_ops()
    if "backbone" in which:
        gen_backbone()
    if "nets" in which:
        gen_nets()
    if "psmnet" in which:
        gen_psmnet()
    if "dsnet" in which:
        gen_dsnet()
    if "hanet" in which:
        gen_hanet()
    if "metrics" in which:
        gen_metrics()
    if "keys" in which:
        gen_keys()
    if "data" in which:
        gen_data()
