"""Data-parallel exchange (sync-BN statistics + flat gradient all-reduce): world_size-2 gloo runs on CPU.
The kernels themselves need a GPU; what is checked here is the protocol and its maths: exchanging
(sum x, sum x^2) forward and (dscale, dshift) backward reproduces single-process BatchNorm on the concatenated
batch, with dgamma/dbeta left as local sums for the gradient all-reduce."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _bn_sync_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import parallel
    parallel.configure(dist.group.WORLD, world)
    torch.manual_seed(0)
    B, C, H, W = 4, 6, 5, 7
    x_full = torch.randn(B, C, H, W, dtype=torch.float64)
    gy_full = torch.randn(B, C, H, W, dtype=torch.float64)
    gamma = torch.rand(C, dtype=torch.float64) + 0.5
    beta = torch.randn(C, dtype=torch.float64)
    eps = 1e-5
    # single-process reference on the concatenated batch
    xr = x_full.clone().requires_grad_(True)
    g_r, b_r = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    yr = torch.nn.functional.batch_norm(xr, None, None, g_r, b_r, True, 0.1, eps)
    yr.backward(gy_full)
    # this rank's shard
    sl = slice(rank * B // world, (rank + 1) * B // world)
    x, gy = x_full[sl], gy_full[sl]
    n_local = x.numel() // C
    sums = torch.stack((x.sum((0, 2, 3)), (x * x).sum((0, 2, 3))))          # what the conv epilogue produces
    parallel.all_reduce_sum_(sums)
    n = parallel.global_count(n_local)
    scale, shift, mean, var = parallel.bn_scale_shift_from_sums(sums[0], sums[1], n, gamma, beta, eps)
    y = x * scale.view(1, C, 1, 1) + shift.view(1, C, 1, 1)
    ok_fwd = torch.allclose(y, yr[sl].detach(), atol=1e-10)
    # backward: local sums -> dgamma/dbeta (local), global sums -> statistics gradient
    invstd = (var + eps).rsqrt()
    ds_l, dh_l = (gy * x).sum((0, 2, 3)), gy.sum((0, 2, 3))
    dgamma_l, dbeta_l = invstd * (ds_l - mean * dh_l), dh_l
    glob = torch.stack((ds_l, dh_l))
    parallel.all_reduce_sum_(glob)
    t = glob[0] - mean * glob[1]
    dinv = gamma * t
    dvar = -0.5 * dinv * invstd ** 3
    dmu = -gamma * invstd * glob[1] - 2 * mean * dvar
    gx = gy * scale.view(1, C, 1, 1) + (dmu / n).view(1, C, 1, 1) + 2 * x * (dvar / n).view(1, C, 1, 1)
    ok_gx = torch.allclose(gx, xr.grad[sl], atol=1e-10)
    # parameter gradients: local pieces summed by the flat gradient all-reduce
    flat = torch.cat((dgamma_l, dbeta_l))
    parallel.all_reduce_sum_(flat)
    ok_p = torch.allclose(flat[:C], g_r.grad, atol=1e-10) and torch.allclose(flat[C:], b_r.grad, atol=1e-10)
    q.put((rank, ok_fwd, ok_gx, ok_p))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_sync_bn_protocol_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_bn_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(r[0] for r in res) == [0, 1]
    for r in res:
        assert r[1] and r[2] and r[3], r


def test_single_rank_is_a_noop():
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import parallel
    parallel.configure(None, 1)
    t = torch.arange(4.0)
    assert parallel.all_reduce_sum_(t) is t and parallel.global_count(7) == 7
