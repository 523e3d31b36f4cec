"""2 real ranks (gloo, one GPU) vs 1 rank: forward outputs per shard, then gradient w.r.t. the network outputs path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import torch.distributed as dist
import torch.multiprocessing as mp

def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from oracle import ref_models as R
    from oracle.detweights import fill_state_dict
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N, parallel, ops
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import synthetic_batch
    parallel.configure(dist.group.WORLD, world)
    calls = {"n": 0}
    orig = parallel.all_reduce_sum_
    def counted(t):
        calls["n"] += 1
        if os.environ.get("DIAG_HOST_REDUCE"):   # take gloo's CUDA path out of the picture: reduce a host copy
            torch.cuda.synchronize()
            h = t.detach().cpu()
            dist.all_reduce(h)
            t.copy_(h)
            torch.cuda.synchronize()
            return t
        return orig(t)
    parallel.all_reduce_sum_ = counted
    m = fill_state_dict(N.minidsnetExt(R.CFG(aspp=0), labels=2, patch_type='1dcorr'), 51).cuda().train()
    full = synthetic_batch(4, 256, 256, seed=77)
    sh = [t[rank * 2:(rank + 1) * 2].contiguous() for t in full]
    acts = {}
    def mk(name):
        def hook(mod, inp, out):
            if isinstance(out, (tuple, list)):
                for i, o in enumerate(out):
                    if torch.is_tensor(o): acts["%s[%d]" % (name, i)] = o.detach().float().cpu().numpy()
            elif torch.is_tensor(out): acts[name] = out.detach().float().cpu().numpy()
        return hook
    for name, mod in m.named_modules():
        if name.count(".") <= 2 and name: mod.register_forward_hook(mk(name))
    outs = m(sh[0], sh[1])
    nf = calls["n"]
    loss = ops.train_loss(outs[0], outs[1], outs[2], sh[2], sh[3], False)
    loss.backward()
    torch.cuda.synchronize()
    g = {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters() if p.grad is not None}
    q.put((rank, [o.detach().float().cpu().numpy() for o in outs[:3]], nf, calls["n"] - nf, g, acts))
    dist.destroy_process_group()

if __name__ == "__main__":
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, 29633, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=500) for _ in procs], key=lambda r: r[0])
    [p.join(60) for p in procs]
    from oracle import ref_models as R
    from oracle.detweights import fill_state_dict
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N, ops
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import synthetic_batch
    m = fill_state_dict(N.minidsnetExt(R.CFG(aspp=0), labels=2, patch_type='1dcorr'), 51).cuda().train()
    full = synthetic_batch(4, 256, 256, seed=77)
    acts = {}
    def mk(name):
        def hook(mod, inp, out):
            if isinstance(out, (tuple, list)):
                for i, o in enumerate(out):
                    if torch.is_tensor(o): acts["%s[%d]" % (name, i)] = o.detach().float().cpu().numpy()
            elif torch.is_tensor(out): acts[name] = out.detach().float().cpu().numpy()
        return hook
    for name, mod in m.named_modules():
        if name.count(".") <= 2 and name: mod.register_forward_hook(mk(name))
    outs = m(full[0], full[1])
    loss = ops.train_loss(outs[0], outs[1], outs[2], full[2], full[3], False)
    r0 = res[0]
    print("first divergences (rank 0 shard vs single-process slice):")
    for name, a in r0[5].items():
        ref = acts.get(name)
        if ref is None: continue
        nb = a.shape[0]
        if ref.shape[0] == 2 * nb and nb == 2: refs = ref[0:2]                 # decoder tensors: batch 4 -> rank 0 holds 0:2
        elif ref.shape[0] == 2 * nb and nb == 4: refs = np.concatenate([ref[0:2], ref[4:6]])   # tower tensors [L0-3,R0-3] -> [L0,L1,R0,R1]
        else: continue
        if refs.shape != a.shape: continue
        print("   %-60s max abs diff %.3e  scale %.2f" % (name, np.abs(a - refs).max(), np.abs(refs).max()))
    loss.backward()
    for r in res:
        print("rank", r[0], "collectives fwd", r[2], "bwd", r[3])
        for i, name in enumerate(("seg1", "disp", "seg2")):
            ref = outs[i][r[0] * 2:(r[0] + 1) * 2].detach().float().cpu().numpy()
            print("   %s max abs diff %.3e (scale %.2f)" % (name, np.abs(r[1][i] - ref).max(), np.abs(ref).max()))
    # gradients: (g_rank0 + g_rank1)/2 vs single
    tot = 0; num = 0
    worst = []
    for k, p in m.named_parameters():
        if p.grad is None or k not in res[0][4]: continue
        ga = 0.5 * (res[0][4][k] + res[1][4][k]); gs = p.grad.cpu().numpy()
        d = np.linalg.norm(ga - gs); n = np.linalg.norm(gs)
        worst.append((d / max(n, 1e-20), k))
    worst.sort()
    print("median rel", worst[len(worst) // 2], "max", worst[-1])
