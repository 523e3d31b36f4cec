"""Training step of the hot path: forward + loss + backward + Adam, optionally captured as ONE hipGraph.

Counterpart of train_model's inner loop (torch_implementation.py:350-397) without its host-side metric /
JPEG work: model(left, right) -> CE(seg1) + CE(seg2) + Lovasz(seg2) + L1(disp) -> backward -> Adam.
Parameters live in one flat f32 buffer (one fused Adam launch, one gradient all-reduce over RCCL when
data-parallel); activations run in `dtype` (bf16 MFMA path or exact f32 path).
"""
import os
import torch

from . import _lib, ops, parallel
from ._lib import call, ptr, stream_ptr


def flatten_parameters(model):
    """Re-home every parameter into one contiguous f32 buffer (and its .grad into a second one)."""
    params = [p for p in model.parameters()]
    n = sum(((p.numel() + 3) // 4) * 4 for p in params)     # 16-byte aligned slices
    dev = params[0].device
    flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
    flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
    off = 0
    with torch.no_grad():
        for p in params:
            k = p.numel()
            flat_p[off:off + k].copy_(p.detach().reshape(-1))
            p.data = flat_p[off:off + k].view(p.shape)
            p.grad = flat_g[off:off + k].view(p.shape)
            off += ((k + 3) // 4) * 4
    return flat_p, flat_g


class TrainStep:
    def __init__(self, model, dtype=torch.bfloat16, lr=0.0015, betas=(0.9, 0.999), eps=1e-7, use_lovasz=True,
                 use_graph=True, world_size=1, process_group=None, use_side_stream=None, loss_fn=None, metrics=None, accumulate=1):
        self.model, self.dtype, self.use_lovasz = model, dtype, use_lovasz
        # metrics.StepMetrics (or None): scored inside the step on the second segmentation head and the disparity, as the
        # reference's lossSeg_fn(seg2) / lossDisp_fn calls do (torch_implementation.py:293,304) — one extra launch, graph-safe
        self.metrics = metrics
        self.loss_fn = loss_fn      # (outputs, seg, disp) -> scalar; default: the joint seg+disp loss of the reference step
        self.lr, self.betas, self.eps = lr, betas, eps
        # gradient accumulation (`-acmt_grad K`, torch_implementation.py:335,362,390-397): K calls form one optimizer step — the
        # gradients of K micro-batches add up in the flat buffer, the K-th call all-reduces, steps with their mean and clears
        self.accumulate = max(1, int(accumulate))
        self._micro = 0
        self.graphs = {}            # (first, last) call of an accumulation cycle -> (hipGraph, its loss tensor)
        self.world_size, self.pg = world_size, process_group
        parallel.configure(process_group, world_size)     # sync-BN statistics exchange + gradient all-reduce
        self.flat_p, self.flat_g = flatten_parameters(model)
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.beta_pow = torch.ones(2, dtype=torch.float32, device=self.flat_p.device)
        self.use_graph = use_graph
        if use_graph and world_size > 1 and _backend_of(process_group) != "nccl":
            # host-staged collectives (gloo: rehearsals and tests on fewer GPUs than ranks) cannot be part of a hipGraph
            import sys
            sys.stderr.write("[TrainStep] process group backend %r cannot be captured: running WITHOUT a graph\n" % _backend_of(process_group))
            self.use_graph = False
        # Weight gradients on a second captured stream: on one GPU every kernel fills the chip and the overlap buys nothing
        # (measured 31.8 ms without vs 32.5 ms with), but under data parallelism the main stream sits in ~400 latency-bound
        # sync-BN all-reduces per step — the weight gradients then run inside those waits.
        # Single GPU, opt-in (SDHIP_TUNE_WGRAD_OVERLAP=1): the side stream carries the decoder's grouped weight gradients, in
        # grids limited to part of the chip, beside the DenseNet backward chain (StepContext.overlap_point).  Measured on
        # one box (ms/step): off 21.29 / 21.31; grids of 64 / 128 / 192 / 256 workgroups 27.5 / 21.9 / 21.5 / 21.7 — the chain's
        # small latency-bound kernels lose what the overlap hides (~2.8 ms slower beside the streaming grids), as in round 2.
        self.overlap_wgrad = world_size == 1 and use_side_stream is None and _lib.TUNE_WGRAD_OVERLAP and not _lib.DIAG_NO_WGRAD_GROUP
        if use_side_stream is None:
            use_side_stream = world_size > 1 or self.overlap_wgrad
        self.use_side_stream = use_side_stream and not _lib.DIAG_NO_SIDE   # timing diagnostics only
        self.graph = None
        self.static = None
        self.loss = None
        # per-step services (ops.StepContext): the first eager step measures / records, later steps use them
        self.ctx = ops.StepContext(self.flat_p.device)
        self.ctx.nbt = []
        self.pack_desc = None
        self.steps_done = 0       # optimizer steps actually EXECUTED (eager steps + graph replays; the recording pass of a capture runs nothing)
        self._capture_fault = None   # tests: a callable invoked inside the capture to make it fail
        self.debug_graph = False     # tests: keep the captured hipGraph inspectable (_lib.graph_node_counts)
        # dropout streams differ per rank (the reference's ranks draw from independently seeded generators) and advance
        # once per step on the device, so that graph replays see new masks (ops.rng_seed_tensor)
        # (the tensor belongs to this step's context: creating a second TrainStep does not restart the first one's stream)
        self.ctx.seed = torch.full((1,), ops.rng_seed_value(_rank_of(process_group) if world_size > 1 else 0), dtype=torch.int64,
                                   device=self.flat_p.device)

    # -- pieces ---------------------------------------------------------------------------------
    def _arm(self):
        """After the first (measuring) step: allocate the zero arena, freeze the weight packs into one batched
        launch, switch parameter gradients to direct accumulation into the flat buffer."""
        self.ctx.allocate_arena()
        rows = ops.pack_descriptors(self.dtype, owners=self.model.parameters())   # this model's weights only
        # The table is replayed for the lifetime of the step (inside the hipGraph): every source must be a slice of THIS
        # step's flat parameter buffer — a row of anything else would read freed memory once its owner is gone (the fault
        # of round 2, tests/test_train.py::test_pack_table_holds_only_the_steps_own_weights)
        lo, hi = self.flat_p.data_ptr(), self.flat_p.data_ptr() + 4 * self.flat_p.numel()
        for r in rows:
            if not (lo <= r[0] < hi):
                raise _lib.SdhipError("weight-pack table holds a source outside this step's parameter buffer")
        self.pack_desc = torch.tensor(rows, dtype=torch.int64, device=self.flat_p.device) if rows else None
        self.ctx.frozen_pack = self.pack_desc is not None
        self.ctx.direct_grads = True
        if self.use_side_stream:
            self.ctx.side = torch.cuda.Stream()
            self.ctx.overlap = self.overlap_wgrad
        self.nbt_tensors = [t for t, _ in self.ctx.nbt]
        self.nbt_incs = [int(i) for _, i in self.ctx.nbt]

    def pack_all(self):
        if self.ctx.frozen_pack:
            call("sdhip_conv_pack_batch", ptr(self.pack_desc), self.pack_desc.shape[0],
                 _lib.BF16 if self.dtype == torch.bfloat16 else _lib.F32, stream_ptr())

    def forward_backward(self, left, right, seg, disp, first=True):
        """first: this call opens an accumulation cycle — the flat gradient buffer is cleared; later calls of the cycle add."""
        ops.set_step_context(self.ctx)
        self.ctx.begin_step()       # one memset clears every zero-initialised workspace of the step
        if first:
            self.flat_g.zero_()
        self.pack_all()             # one launch packs every weight (forward and data-grad orientation)
        outs = self.model(left.to(self.dtype), right.to(self.dtype))
        if self.loss_fn is not None:
            loss = self.loss_fn(outs, seg, disp)
        else:
            loss = ops.train_loss(outs[0], outs[1], outs[2], seg, disp, self.use_lovasz)
        if self.metrics is not None and self.loss_fn is None:
            self.metrics.update(outs[2].detach(), seg, outs[1].detach(), disp)
        loss.backward()
        self.ctx.join()             # weight gradients ran on the side stream
        return loss.detach()

    def all_reduce(self):
        parallel.all_reduce_sum_(self.flat_g)   # one RCCL sum over xGMI per step; Adam divides by world_size

    def optimizer_step(self):
        call("sdhip_adam_step", ptr(self.flat_p), ptr(self.flat_g), ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(self.beta_pow),
             self.flat_p.numel(), self.lr, self.betas[0], self.betas[1], self.eps, 0.0, 1.0 / (self.world_size * self.accumulate),
             stream_ptr())
        ops.invalidate_packed_weights()

    def _phase(self):
        """(first, last) of the call about to run within its accumulation cycle; advances the cycle counter."""
        first = self._micro == 0
        self._micro += 1
        last = self._micro >= self.accumulate
        if last:
            self._micro = 0
        return first, last

    def _eager(self, left, right, seg, disp, phase=None):
        """One (micro-)step.  phase = (first, last) of the accumulation cycle (None: decided by the call counter)."""
        first, last = phase if phase is not None else self._phase()
        loss = self.forward_backward(left, right, seg, disp, first)
        if last:
            self.all_reduce()
            self.optimizer_step()
        if self.ctx.arena is None:
            self._arm()
        elif self.nbt_tensors:
            torch._foreach_add_(self.nbt_tensors, self.nbt_incs)   # BatchNorm.num_batches_tracked, one launch
        # next step draws new dropout masks.  A device-side add AFTER the backward pass (which regenerates this step's
        # masks from the same seed): captured into the graph, so every replay advances it too.
        self.ctx.seed.add_(1)
        if last and not torch.cuda.is_current_stream_capturing():
            self.steps_done += 1
        return loss

    # -- public ---------------------------------------------------------------------------------
    def capture(self, left, right, seg, disp, warmup=2):
        """Warm up eagerly on a side stream, then record forward+backward(+all-reduce)+Adam as one hipGraph."""
        self.static = [t.clone() for t in (left, right, seg, disp)]
        # warm-up and capture share ONE side stream: autograd's AccumulateGrad nodes remember the stream they were created
        # on (the first forward), and a different capture stream would make every parameter gradient cross streams
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            for _ in range(warmup * self.accumulate - self._micro):      # whole accumulation cycles: the capture starts at a cycle boundary
                self._eager(*self.static)
        torch.cuda.current_stream().wait_stream(cap)
        torch.cuda.synchronize()
        # No device memory may be released while the stream records (what `torch.cuda.graph` guards against the same way):
        # collect the garbage of whatever ran before NOW, and keep the cyclic collector off until the capture has ended —
        # a discarded model collected in the middle of the recording pass frees tensors, fires the weight-cache callbacks
        # and, after an injected failure, took the process down (tests/test_train.py in the middle of the full suite).
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        # data parallel: the RCCL watchdog thread polls events while we capture; thread-local capture errors keep its
        # (legal) calls from invalidating the capture of this thread
        mode = "thread_local" if self.world_size > 1 else "global"
        # The capture is driven by hand instead of through `torch.cuda.graph`: that context manager re-raises out of its
        # __exit__ when the capture was invalidated and then neither restores the current stream nor ends the capture,
        # which leaves the whole process unable to launch (observed on ROCm 7.2: every later call fails with
        # hipErrorStreamCaptureInvalidated).  Here a failure ends the capture explicitly (sdhip_abort_capture).
        gc_was_on = gc.isenabled()
        gc.disable()
        self._gen_state = self._torch_generator_state()    # restored exactly should the capture fail (_abandon_capture)
        graphs = {}
        K = self.accumulate
        phases = [(True, True)] if K == 1 else ([(True, False), (False, True)] if K == 2 else [(True, False), (False, False), (False, True)])
        try:
            # one graph per kind of call of an accumulation cycle: the opening one clears the gradient buffer, the closing one
            # holds the all-reduce and Adam (accumulate = 1: one graph that does both)
            for ph in phases:
                graph = torch.cuda.CUDAGraph(keep_graph=self.debug_graph)    # kept: _lib.graph_node_counts (node inventory tests)
                with torch.cuda.stream(cap):
                    graph.capture_begin(capture_error_mode=mode)
                    try:
                        loss = self._eager(*self.static, phase=ph)
                        if self._capture_fault is not None:
                            self._capture_fault()
                        graph.capture_end()
                    except BaseException:
                        self._close_broken_capture(graph, cap)
                        raise
                graphs[ph] = (graph, loss)
        finally:
            if gc_was_on:
                gc.enable()
        torch.cuda.current_stream().wait_stream(cap)
        self.graphs = graphs
        self.graph, self.loss = graphs[phases[-1]]      # the closing call's graph (accumulate = 1: the only one)
        return self

    def _close_broken_capture(self, graph, cap):
        """Best-effort return to a launchable process after an exception inside the capture."""
        try:
            graph.capture_end()                  # ends the allocator's pool redirection when the capture itself is intact
        except BaseException:
            pass
        streams = [cap] + ([self.ctx.side] if self.ctx.side is not None else [])
        for st in streams:
            try:
                _lib.abort_capture(st)
            except _lib.SdhipError as e:
                import sys
                sys.stderr.write("[TrainStep] %s\n" % e)
        try:
            torch._C._cuda_endAllocateToPool(self.flat_p.device.index or 0, graph.pool())
        except BaseException:
            pass
        del graph

    def _torch_generator_state(self):
        try:
            dev = self.flat_p.device
            return torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()].get_state()
        except BaseException:
            return None

    def _abandon_capture(self):
        """A capture that raised recorded launches but executed none: device state (parameters, moments, running
        statistics, dropout seed) is that of the last eager warm-up step.  Only host-side bookkeeping of the half-recorded
        step has to be reset before eager steps continue."""
        self.graph, self.graphs, self.use_graph, self.loss = None, {}, False, None
        torch.cuda.synchronize()               # raises if the process could not be brought back: nothing can run then
        # capture_begin put torch's default CUDA generator into capture mode and only a completed capture_end takes it out
        # again ("Offset increment outside graph capture" on the next torch.randn(device='cuda')): give the generator a
        # fresh, non-capturing state object carrying EXACTLY the seed and offset it had before the capture began — later
        # torch CUDA draws continue the stream instead of replaying it from its start
        try:
            dev = self.flat_p.device
            gen = torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
            fresh = torch.Generator(device=dev)
            if getattr(self, "_gen_state", None) is not None:
                fresh.set_state(self._gen_state)
            else:
                fresh.manual_seed(gen.initial_seed())
            gen.graphsafe_set_state(fresh.graphsafe_get_state())
        except BaseException as e:
            import sys
            sys.stderr.write("[TrainStep] could not reset the capture state of torch's CUDA generator: %s\n" % str(e).splitlines()[0])
        self.ctx.unpacks = []
        self.ctx.wq = []
        self.ctx.wq_keep.clear()
        self.ctx.keep.clear()
        ops.invalidate_packed_weights()

    def __call__(self, left, right, seg, disp):
        if not self.use_graph:
            return self._eager(left, right, seg, disp)
        if self.graph is None:
            try:
                self.capture(left, right, seg, disp)
            except RuntimeError as e:   # e.g. a collective that cannot be captured on this RCCL build: run eagerly, loudly
                import sys
                sys.stderr.write("[TrainStep] hipGraph capture failed (%s); continuing WITHOUT a graph\n" % str(e).splitlines()[0])
                self._abandon_capture()
                return self._eager(left, right, seg, disp)
        for dst, src in zip(self.static, (left, right, seg, disp)):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        ph = self._phase()
        graph, loss = self.graphs[ph]
        graph.replay()
        if ph[1]:
            self.steps_done += 1
        return loss


def _backend_of(pg):
    import torch.distributed as dist
    try:
        return dist.get_backend(pg)
    except Exception:
        return None


def _rank_of(pg):
    import torch.distributed as dist
    return dist.get_rank(pg) if dist.is_available() and dist.is_initialized() else 0


def synthetic_batch(B, H, W, labels=2, device="cuda", seed=1234):
    """S-ROSeS-shaped synthetic batch (tensor contract of util/utilTorchDataLoader.py:176-258,608-630)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    left = torch.rand((B, 3, H, W), generator=g)
    right = torch.rand((B, 3, H, W), generator=g)
    lab = torch.randint(0, labels, (B, H, W), generator=g)
    seg = torch.nn.functional.one_hot(lab, labels).permute(0, 3, 1, 2).float().contiguous()
    disp = (torch.rand((B, 1, H, W), generator=g) * 8.0 * seg[:, 1:2] + 0.1).contiguous()
    return [t.to(device) for t in (left, right, seg, disp)]
