#!/usr/bin/env python
"""Headline benchmark: stereo-pairs/s of one training step (forward + loss + backward + Adam) of the joint
segmentation + disparity network on synthetic S-ROSeS-shaped 512x256 (WxH) batches.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL; per-GPU batch fixed => weak scaling)

Prints ONE JSON line on rank 0 (contract in the round prompt) including
  roofline     : the dominant kernel (5x5 64->64 conv of Conv2DownUp5) timed with HIP events on its own stream,
                 algorithmic FLOPs / launch time vs the dense bf16 MFMA peak;
  cpu_baseline : the CPU oracle (oracle/ref_models.py, "port") timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


class _LazyTorch:
    """torch on first use: the parent of a multi-rank launch must stay free of torch / HIP (spawn_ranks), while the helper
    functions below (and the tools that import this module for them) see an ordinary module."""

    def __getattr__(self, name):
        import torch as t
        globals()["torch"] = t
        return getattr(t, name)


torch = _LazyTorch()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="stereo pairs per GPU (shipped recipe: -b 8)")
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--model", default="minidsnetExt", choices=["minidsnetExt", "psmnet", "dsnet", "dsnetnoCorr", "minidsnetExt_cfg5"],
                    help="psmnet = BASELINE config 3 (PSMNet(192), build-defined loss: mean L1 of the three predictions)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--roofline-only", action="store_true",
                    help="run only the dominant-kernel microbenchmark (the command profiled under profiles/: tools/roofline_profile.sh)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (production).  gloo: rehearsal of the N-rank path on a box with fewer GPUs than ranks "
                         "(ranks share devices, collectives go through the host; the number it prints is not a result)")
    ap.add_argument("--rank-probe", action="store_true", help="each rank prints its launch environment and exits (CPU test of the launcher)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary single-GPU results (dsnet = config 2 as literally named, PSMNet(192) B=8 = config 3)")
    ap.add_argument("--secondary-steps", type=int, default=6)
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--cpu-warmup", type=int, default=3)
    ap.add_argument("--cpu-threads", type=int, default=16)
    return ap.parse_args()


def build_model(dtype, name="minidsnetExt"):
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N
    CFG = N.CFG                                # plain attribute bag (the argparse fields the model reads)
    torch.manual_seed(0)
    if name == "dsnet":
        return N.dsnet(CFG(), labels=2, pretrained=False).cuda().train()
    if name == "dsnetnoCorr":
        return N.dsnetnoCorr(CFG(), labels=2, pretrained=False).cuda().train()
    if name == "psmnet":
        from pmt_learning_for_semantic_segmentation_and_disparity_amd.psmnet import PSMNet
        return PSMNet(192).cuda().train()
    if name == "minidsnetExt_cfg5":     # BASELINE config 5: ASPP + HANet, 19 Cityscapes classes (the HANet head is built, and as upstream unused with aspp=2)
        return N.minidsnetExt(CFG(dropout=0.0, aspp=2, use_att=1, hanet=1), labels=19, pretrained=False, patch_type='1dcorr',
                              backbone='densenet').cuda().train()
    m = N.minidsnetExt(CFG(dropout=0.0, aspp=0, use_att=1), labels=2, pretrained=False, patch_type='1dcorr', backbone='densenet')
    return m.cuda().train()


def kernel_roofline(dtype, B, H, W):
    """Time the dominant kernel alone: conv2dSame 5x5 64->64 at full resolution (Conv2DownUp5.c1-c3, 63 % of the FLOPs)."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
    from pmt_learning_for_semantic_segmentation_and_disparity_amd._lib import call, ptr
    C = 64
    x = torch.randn(B, H, W, C, device="cuda").to(dtype).permute(0, 3, 1, 2)
    w = torch.randn(C, C, 5, 5, device="cuda") * 0.03
    wp = ops.packed_weight(w, 'conv', 'fwd', dtype)
    y = ops.empty_nhwc(B, C, H, W, dtype, "cuda")
    st = torch.zeros(ops.NREP, 1, 2, C, dtype=torch.float64, device="cuda")
    s = torch.cuda.Stream()
    n = 20
    with torch.cuda.stream(s):
        def launch():
            ops._conv_launch(x, C, wp, y, C, None, None, None, st, B, H, W, C, H, W, C, 5, 5, 1, 1, 2, 2, False, 1, 0, False, ops.NREP)
        for _ in range(3):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(n):
            launch()
        e1.record(s)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / n
    flops = 2.0 * B * H * W * C * C * 25
    peak = 2500.0 if dtype == torch.bfloat16 else 157.3
    ach = flops / (ms * 1e-3) / 1e12
    # HBM bytes per launch of this kernel from the PMC passes committed under profiles/ (FETCH_SIZE doubled as the gfx950
    # note of MI355X_MICROARCH.md prescribes, + WRITE_SIZE; tools/roofline_profile.sh); null when no profile is committed
    traffic, src, busy = None, None, None
    for name in ("r03_roofline_traffic.json", "r02_roofline_traffic.json", "r01_roofline_traffic.json"):
        tj = os.path.join(ROOT, "profiles", name)
        if dtype == torch.bfloat16 and (B, H, W) == (8, 256, 512) and os.path.exists(tj):
            try:
                j = json.load(open(tj))
                traffic, src, busy = j.get("hbm_bytes_per_launch"), "profiles/" + name, j.get("mfma_busy_frac")
                break
            except Exception:
                pass
    cyc = None
    try:   # (bench shape only) matrix-pipe busy share of the GPU CYCLES of the launch (GRBM_GUI_ACTIVE based; tools/pmc_conv5.sh), committed with the profiles
        if dtype == torch.bfloat16 and (B, H, W) == (8, 256, 512):
            cyc = json.load(open(os.path.join(ROOT, "profiles", "r02_band_counters.json")))["_derived"]["mfma_busy_share_of_gpu_cycles"]
    except Exception:
        pass
    return {"bound": "mfma", "kernel": "conv_band_kernel<5x5, 64 out-ch> (persistent, 16x32 tiles, halo prefetched by LDS-DMA) 64->64 @%dx%dx%d" % (B, H, W),
            "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
            "traffic_source": (src + " (PMC passes of `bench.py --roofline-only`, tools/roofline_profile.sh; not measured in this run)") if src else None,
            "mfma_busy_frac": busy, "mfma_busy_cycle_frac": cyc, "algorithmic_bytes": 2 * B * H * W * C * (2 if dtype == torch.bfloat16 else 4) + C * C * 25 * 2,
            "ms_per_launch": round(ms, 4)}


def hbm_kernel_roofline(dtype, B, H, W):
    """The largest HBM-bound kernel family of the step, timed alone with HIP events on its own stream: the second pass of
    the BatchNorm backward (`bn_bwd_apply_fin`) on a (B,64,H,W) map — reads the incoming gradient and the raw convolution
    output, writes the gradient of the raw output: 3 tensors of B*H*W*64 elements (+ per-channel vectors)."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops, _lib
    from pmt_learning_for_semantic_segmentation_and_disparity_amd._lib import call, ptr
    C = 64
    npix = B * H * W
    g = torch.randn(B, H, W, C, device="cuda").to(dtype)
    x = torch.randn(B, H, W, C, device="cuda").to(dtype)
    gx = torch.empty_like(x)
    vec = lambda v: torch.full((1, C), v, dtype=torch.float32, device="cuda")
    scale, shift, mean, invstd, gamma, beta = vec(1.0), vec(0.1), vec(0.0), vec(1.0), vec(1.0)[0].clone(), vec(0.0)[0].clone()
    dsc = torch.zeros(ops.NREP, 1, C, device="cuda"); dsh = torch.zeros(ops.NREP, 1, C, device="cuda")
    dgamma = torch.zeros(C, device="cuda"); dbeta = torch.zeros(C, device="cuda")
    dt = _lib.BF16 if dtype == torch.bfloat16 else _lib.F32
    s = torch.cuda.Stream()
    n = 20
    with torch.cuda.stream(s):
        def launch():
            call("sdhip_bn_bwd_apply_fin", ptr(g), C, ptr(x), C, ptr(gx), C, ptr(scale), ptr(shift), ptr(dsc), ptr(dsh), ops.NREP,
                 ptr(gamma), ptr(mean), ptr(invstd), ptr(dgamma), ptr(dbeta), 0, 1.0, npix, C, 1, float(npix), 1, dt,
                 ctypes_stream(s))
        for _ in range(3):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(n):
            launch()
        e1.record(s)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / n
    nbytes = 3 * npix * C * (2 if dtype == torch.bfloat16 else 4)
    ach = nbytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "bn_bwd_apply_fin (BatchNorm backward, second pass) on (%d,64,%d,%d)" % (B, H, W),
            "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4),
            "algorithmic_bytes": nbytes, "ms_per_launch": round(ms, 4), "traffic": None}


def ctypes_stream(s):
    import ctypes
    return ctypes.c_void_p(s.cuda_stream)


# algorithmic work per stereo pair of one training step at 256x512 (forward MACs x 2 FLOP x 3 for fwd+dgrad+wgrad;
# conv activation bytes, each conv reading its input and writing its output once, x 3); scaled by H*W
WORK = {"minidsnetExt": (643.0e9, 2.3e9), "dsnet": (3 * 2 * 195.4e9, 2.3e9 * 195.4 / 107.17), "psmnet": (1108.2e9, 3 * 0.73e9 + 3 * 0.4e9),
        "minidsnetExt_cfg5": (3 * 2 * 98.1e9, 2.3e9 * 98.1 / 107.17)}     # aspp=2: 98.1 GMAC forward per pair at 256x512


def step_roofline(model, B, H, W, ms, dtype):
    fl, by = WORK[model]
    k = (H * W) / (256.0 * 512.0)
    fl, by = fl * k * B, by * k * B * (1.0 if dtype == "bf16" else 2.0)
    t = ms * 1e-3
    mf_peak = 2500.0 if dtype == "bf16" else 157.3
    return {"flops_per_step": fl, "bytes_per_step": by, "achieved_tflops": round(fl / t / 1e12, 1), "mfma_peak_tflops": mf_peak,
            "mfma_frac": round(fl / t / 1e12 / mf_peak, 4), "achieved_gbs": round(by / t / 1e9, 1), "hbm_peak_gbs": 8000.0,
            "hbm_frac": round(by / t / 1e9 / 8000.0, 4),
            "ideal_ms": round((fl / (mf_peak * 1e12) + by / 8.0e12) * 1e3, 2),
            "note": "whole training step (fwd + loss + bwd + Adam): algorithmic FLOPs and conv-activation bytes of SURVEY 8d vs both roofs"}


def time_model(name, dtype, B, H, W, steps, warmup, world=1, pg=None, use_graph=True, seed=1234):
    """Build `name`, run `warmup` untimed + `steps` timed training steps on a resident synthetic batch; (ms/step, loss, graph?)."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops as _ops
    model = build_model(dtype, name)
    loss_fn = None
    if name == "psmnet":
        loss_fn = lambda outs, seg, disp: _ops.mean_l1_loss(outs, disp[:, 0])
    elif name in ("dsnet", "dsnetnoCorr"):
        loss_fn = lambda outs, seg, disp: _ops.train_loss(outs[0], outs[1], outs[2], seg, disp, True)
    elif name == "minidsnetExt_cfg5":   # the cityscapes rules: void pixels, disp > 0 mask
        loss_fn = lambda outs, seg, disp: _ops.train_loss(outs[0], outs[1], outs[2], seg, disp, True, True, True)
    step = TrainStep(model, dtype=dtype, use_graph=use_graph, world_size=world, process_group=pg, loss_fn=loss_fn)
    batch = synthetic_batch(B, H, W, labels=19 if name == "minidsnetExt_cfg5" else 2, seed=seed)
    for _ in range(warmup):
        loss = step(*batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step(*batch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = (dt / steps * 1e3, float(loss.item()), step.use_graph)
    _ops.set_step_context(None)
    del step, model, batch
    torch.cuda.empty_cache()
    return out


def cpu_baseline(B, H, W, steps, threads=16, warmup=3):
    """The CPU oracle (a port of the reference graph to plain torch.nn) on the host cores: fwd + loss + bwd."""
    import torch.nn.functional as F
    from oracle import ref_models as R
    from oracle.losses_ref import lovasz_softmax_onehot as _lovasz_softmax_torch
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import synthetic_batch
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(threads, avail))    # the GPU box gives one GPU's share of the host (16 cores), not the whole socket
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = R.minidsnetExt(R.CFG(), labels=2, patch_type='1dcorr').train()
    left, right, seg, disp = synthetic_batch(B, H, W, device="cpu")
    ts = []
    for i in range(steps + warmup):
        sys.stderr.write("[bench] cpu_baseline step %d/%d on %d threads\n" % (i, steps + warmup, cores)); sys.stderr.flush()
        t0 = time.time()
        o = m(left, right)
        ce = lambda y: torch.mean(torch.sum(-seg * F.log_softmax(y, 1), 1))
        loss = ce(o[0]) + ce(o[2]) + _lovasz_softmax_torch(o[2], seg) + F.l1_loss(o[1], disp)
        m.zero_grad(set_to_none=True)
        loss.backward()
        ts.append(time.time() - t0)
    t = sum(ts[warmup:]) / max(1, len(ts) - warmup)
    return {"value": round(B / t, 4), "unit": "stereo-pairs/s", "cores": cores, "kind": "port",
            "sample": "%d steps of B=%d %dx%d fp32 fwd+loss(CE+CE+Lovasz+L1)+bwd after %d warm-up steps (oracle/ref_models.py, the "
                      "CPU restatement pinned to the reference by tests/golden)" % (steps, B, W, H, warmup)}


def spawn_ranks(a):
    """`python bench.py --gpus N` with no launcher environment: the parent starts N fresh rank processes (one per GPU,
    the counterpart of mp.spawn(runNetwork, nprocs) at torch_implementation.py:975) and relays their output.  The parent
    never touches the GPU (no torch import, no HIP call) and exits non-zero when any rank does."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live and not rc:
        time.sleep(0.2)
        for p in list(live):
            if p.poll() is not None:
                live.remove(p)
                rc = rc or p.returncode
    for p in live:   # one rank died: do not leave its peers hanging in a collective
        p.kill()
        p.wait()
    sys.exit(rc)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (a.gpus, world))
    if a.rank_probe:
        print(json.dumps({"rank": rank, "local_rank": local, "world": world, "master": "%s:%s" % (
            os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT")), "pid": os.getpid(), "ppid": os.getppid()}), flush=True)
        return
    ndev = torch.cuda.device_count()
    if a.backend == "nccl" and local >= ndev:
        raise SystemExit("bench.py: rank %d needs GPU %d but this node shows %d (RCCL wants one GPU per rank)" % (rank, local, ndev))
    torch.cuda.set_device(local % max(1, ndev))
    pg = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(a.backend)   # "nccl" IS RCCL on ROCm; MASTER_ADDR/PORT from the launcher env
        pg = dist.group.WORLD
        assert dist.get_world_size() == a.gpus
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    if a.roofline_only:
        import pmt_learning_for_semantic_segmentation_and_disparity_amd  # noqa: F401
        print(json.dumps({"roofline": kernel_roofline(dtype, a.batch, a.height, a.width)}))
        return
    model = build_model(dtype, a.model)
    loss_fn = None
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops as _ops
    if a.model == "psmnet":
        loss_fn = lambda outs, seg, disp: _ops.mean_l1_loss(outs, disp[:, 0])
    elif a.model in ("dsnet", "dsnetnoCorr"):     # log-softmax heads: the same CE + Lovasz + L1 composition applies to them
        loss_fn = lambda outs, seg, disp: _ops.train_loss(outs[0], outs[1], outs[2], seg, disp, True)
    elif a.model == "minidsnetExt_cfg5":          # the cityscapes rules: void pixels, disp > 0 mask
        loss_fn = lambda outs, seg, disp: _ops.train_loss(outs[0], outs[1], outs[2], seg, disp, True, True, True)
    step = TrainStep(model, dtype=dtype, use_graph=not a.no_graph, world_size=world, process_group=pg, loss_fn=loss_fn)
    batch = synthetic_batch(a.batch, a.height, a.width, labels=19 if a.model == "minidsnetExt_cfg5" else 2, seed=1234 + rank)
    for _ in range(a.warmup):
        loss = step(*batch)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step(*batch)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    lossv = float(loss.item())
    if rank == 0:
        out = {"metric": "stereo-pairs/sec (train fwd+bwd) 512x256", "value": round(a.batch * world * a.steps / dt, 3),
               "unit": "stereo-pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": ("minidsnetExt (densenet121, 1dcorr, aspp 0; the live PyTorch form of baseline_SDnet*) "
                                       "train step fwd+loss(CE+CE+Lovasz+L1)+bwd+Adam, %dx%d (WxH), batch %d per GPU, %s"
                                       if a.model == "minidsnetExt" else
                                       "dsnet (PyTorch port of baseline_SDnet_small_fixed, 2-D corr) train step fwd+loss+bwd+Adam, %dx%d (WxH), batch %d per GPU, %s"
                                       if a.model == "dsnet" else
                                       "minidsnetExt(aspp=2, hanet=1, 19 classes; BASELINE config 5) train step fwd+loss(cityscapes rules)+bwd+Adam, %dx%d (WxH), batch %d per GPU, %s"
                                       if a.model == "minidsnetExt_cfg5" else
                                       "dsnetnoCorr (baseline_SDnet_small) train step fwd+loss+bwd+Adam, %dx%d (WxH), batch %d per GPU, %s"
                                       if a.model == "dsnetnoCorr" else
                                       "PSMNet(192) stacked hourglass train step fwd+loss(mean L1 x3)+bwd+Adam, %dx%d (WxH), batch %d per GPU, %s")
                                      % (a.width, a.height, a.batch, "hipGraph" if step.use_graph else "eager"),
                          "global_batch": a.batch * world, "parallelism": "dp%d" % world,
                          "world_size": (torch.distributed.get_world_size() if world > 1 else 1),
                          "backend": ("rccl" if a.backend == "nccl" else "gloo (rehearsal, not a result)") if world > 1 else None},
               "loss": round(lossv, 5)}
        if world > 1:
            # from the builder's side (one GPU per lease) the N-rank RCCL step had never executed before this run;
            # the line says what actually ran so that a silently degraded path cannot pass for the production one
            out["multi_rank_path"] = {"graph_with_collectives": bool(step.use_graph), "backend": a.backend,
                                      "note": "first hardware execution of the RCCL path happens in the driver's scaling run; "
                                              "builder-side coverage: 2-rank gloo (CPU + 1 GPU), 1-rank RCCL capture"}
        sys.stderr.write("[bench] timed region done: %.3f ms/step\n" % (dt / a.steps * 1e3)); sys.stderr.flush()
        out["step_roofline"] = step_roofline(a.model if a.model != "dsnetnoCorr" else "dsnet", a.batch * world, a.height, a.width,
                                             dt / a.steps * 1e3, a.dtype)
        if not a.no_roofline:
            del step, model
            _ops.set_step_context(None)
            torch.cuda.empty_cache()
            out["roofline"] = kernel_roofline(dtype, a.batch, a.height, a.width)
            out["roofline_hbm"] = hbm_kernel_roofline(dtype, a.batch, a.height, a.width)
        if world == 1 and not a.no_secondary and a.model == "minidsnetExt" and not a.no_graph:
            sec = []
            f32 = torch.float32
            for name, sb, sdt, sh, sw, label in (
                    ("dsnet", a.batch, dtype, a.height, a.width, "dsnet = PyTorch port of baseline_SDnet_small_fixed (BASELINE config 2 as literally named)"),
                    ("psmnet", 8, dtype, a.height, a.width, "PSMNet(192) stacked hourglass, loss mean L1 x3 (BASELINE config 3, SURVEY 8d batch 8)"),
                    ("minidsnetExt", a.batch, f32, a.height, a.width, "minidsnetExt, the headline workload on the fp32 path (the arithmetic north_star's 1e-3 "
                                                                     "parity gate is stated for: f32 activations, v_mfma_f32_16x16x4_f32)"),
                    ("psmnet", 4, dtype, 512, 960, "PSMNet(192) at BASELINE config 4's per-GPU workload (960x512, batch 4 per GPU) on ONE GPU"),
                    ("minidsnetExt_cfg5", 4, dtype, 512, 1024, "minidsnetExt(aspp=2, hanet=1, 19 classes) at BASELINE config 5's image size (1024x512), "
                                                               "batch 4, cityscapes loss rules, on ONE GPU")):
                sname = "f32" if sdt == f32 else a.dtype
                sys.stderr.write("[bench] secondary: %s B=%d %s\n" % (name, sb, sname)); sys.stderr.flush()
                ms, lv, gr = time_model(name, sdt, sb, sh, sw, a.secondary_steps, 1)
                sec.append({"workload": "%s, %dx%d, batch %d, %s, %s" % (label, sw, sh, sb, sname, "hipGraph" if gr else "eager"),
                            "dtype": sname, "value": round(sb / ms * 1e3, 2), "unit": "stereo-pairs/s", "ms_per_step": round(ms, 3),
                            "steps": a.secondary_steps, "loss": round(lv, 5), "step_roofline": step_roofline(name, sb, sh, sw, ms, sname)})
            out["secondary"] = sec
        if world == 1 and not a.no_cpu_baseline and a.model == "minidsnetExt":
            out["cpu_baseline"] = cpu_baseline(2, a.height, a.width, a.cpu_steps, a.cpu_threads, a.cpu_warmup)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()     # rank 0 may still be in its single-rank roofline measurement: leave together
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
