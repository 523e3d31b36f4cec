"""One conv+BN+ReLU layer: 2 real ranks (gloo) vs 1 rank on the joint batch — output and gradient differences."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import torch.distributed as dist
import torch.multiprocessing as mp

def build():
    from oracle.detweights import fill_state_dict
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N
    return fill_state_dict(N.Conv2DownUp(8, 16, 3, True), 61).cuda().train()

def data():
    from oracle.detweights import randn_input
    return randn_input(61, "x", (4, 8, 64, 96)), randn_input(62, "g", (4, 16, 64, 96))

def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import parallel
    parallel.configure(dist.group.WORLD, world)
    m = build(); x, g = data()
    xs = x[rank * 2:(rank + 1) * 2].cuda().requires_grad_(True)
    y = m(xs)
    y.backward(g[rank * 2:(rank + 1) * 2].cuda())
    torch.cuda.synchronize()
    q.put((rank, y.detach().cpu().numpy(), xs.grad.cpu().numpy(), {k: p.grad.cpu().numpy() for k, p in m.named_parameters()}))
    dist.destroy_process_group()

if __name__ == "__main__":
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, 29644, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    [p.join(60) for p in procs]
    m = build(); x, g = data()
    xs = x.cuda().requires_grad_(True)
    y = m(xs); y.backward(g.cuda())
    yr = y.detach().cpu().numpy(); gx = xs.grad.cpu().numpy()
    for r in res:
        sl = slice(r[0] * 2, r[0] * 2 + 2)
        print("rank", r[0], "y max diff %.3e (scale %.2f)  gx max diff %.3e (scale %.2f)" % (np.abs(r[1] - yr[sl]).max(), np.abs(yr).max(), np.abs(r[2] - gx[sl]).max(), np.abs(gx).max()))
    for k, p in m.named_parameters():
        gs = p.grad.cpu().numpy(); ga = res[0][3][k] + res[1][3][k]
        print("   %-34s rel %.3e" % (k, np.linalg.norm(ga - gs) / max(np.linalg.norm(gs), 1e-20)))
