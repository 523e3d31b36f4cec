"""Autograd bindings of the libsdhip.so kernels.

Tensors keep the reference's logical NCHW shapes but live in channels-last
(NHWC) memory, possibly as a channel slice of a wider slab; `nhwc_view`
returns the pixel stride the C ABI wants.  Nothing in this file computes on the
CPU or through ATen kernels: a non-GPU tensor is an error.
"""
import os
import torch

from . import _lib
from ._lib import call, dtype_code, ptr, stream_ptr


def _require_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.SdhipError("sdhip ops need GPU tensors (got %s); there is no CPU path" % t.device)


def nhwc_view(x):
    """Return (tensor, ld): x itself if its memory is NHWC with a uniform pixel stride, else a channels-last copy."""
    B, C, H, W = x.shape
    sb, sc, sh, sw = x.stride()
    ld = sw
    ok = (sc == 1 or C == 1) and ld >= C and (sh == W * ld or H == 1) and (sb == H * W * ld or B == 1)
    if W == 1:
        ok = False if (sc != 1 and C != 1) else (sh >= C and (sb == H * sh or B == 1))
        ld = sh if ok else ld
    if not ok:
        x = x.contiguous(memory_format=torch.channels_last)
        if x.stride(1) != 1 and C != 1:  # degenerate shapes: force an explicit NHWC buffer
            x = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        ld = C
    return x, ld


def empty_nhwc(B, C, H, W, dtype, device):
    return torch.empty((B, H, W, C), dtype=dtype, device=device).permute(0, 3, 1, 2)


def alloc_nhwc(B, C, H, W, dtype, device):
    """(tensor, ld): NHWC tensor whose pixel stride is rounded up to 8 elements (16 bytes of bf16), so that tensors
    with odd channel counts (the 65-channel concats, 1-channel maps of models/dsnet_t2.py:1128-1170) still take the
    16-byte vector / LDS-DMA paths of the conv kernels.  The pad channels are never read as data (the kernels mask
    the channel tail) and never written."""
    ld = (C + 7) & ~7
    if ld == C:
        return empty_nhwc(B, C, H, W, dtype, device), C
    return torch.empty((B, H, W, ld), dtype=dtype, device=device)[..., :C].permute(0, 3, 1, 2), ld


def aligned_view(x):
    """nhwc_view + 16-byte-aligned pixels: a tensor whose pixel stride is not a multiple of 8 elements (1- or 2-channel
    maps coming from outside the conv stack, e.g. the loss gradients) is copied once into a padded buffer, so that the
    conv kernels read it with 16-byte vectors instead of element by element."""
    xv, ld = nhwc_view(x)
    if ld % 8 == 0:
        return xv, ld
    B, C, H, W = xv.shape
    y, ldy = alloc_nhwc(B, C, H, W, xv.dtype, xv.device)
    call("sdhip_affine_act", ptr(xv), ld, ptr(y), ldy, None, 0, None, None, B * H * W, C, 1, 0, dtype_code(xv), stream_ptr())
    return y, ldy


class _CorrFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, in1, in2, PH, PW, dil):
        _require_gpu(in1, in2)
        if in1.shape != in2.shape or in1.dtype != in2.dtype:
            raise _lib.SdhipError("correlation inputs must have the same shape and dtype")
        B, C, H, W = in1.shape
        a, lda = nhwc_view(in1)
        b, ldb = nhwc_view(in2)
        if lda != ldb:
            a = a.contiguous(memory_format=torch.channels_last); b = b.contiguous(memory_format=torch.channels_last)
            lda = ldb = C
        out = torch.empty((B, H, W, PH, PW), dtype=in1.dtype, device=in1.device)
        call("sdhip_corr_fwd", ptr(a), ptr(b), ptr(out), B, H, W, C, lda, PH, PW, dil, PH * PW,
             dtype_code(in1), stream_ptr())
        ctx.save_for_backward(a, b)
        ctx.cfg = (PH, PW, dil, lda)
        # (B,PH,PW,H,W) view of NHWC memory, displacement fastest
        return out.permute(0, 3, 4, 1, 2)

    @staticmethod
    def backward(ctx, gout):
        a, b = ctx.saved_tensors
        PH, PW, dil, ld = ctx.cfg
        B, C, H, W = a.shape
        g = gout.permute(0, 3, 4, 1, 2)  # (B,H,W,PH,PW)
        if not g.is_contiguous():
            g = g.contiguous()
        if ld != C:
            a = a.contiguous(memory_format=torch.channels_last); b = b.contiguous(memory_format=torch.channels_last)
            ld = C
        ga = empty_nhwc(B, C, H, W, a.dtype, a.device)
        gb = empty_nhwc(B, C, H, W, a.dtype, a.device)
        call("sdhip_corr_bwd", ptr(a), ptr(b), ptr(g), ptr(ga), ptr(gb), B, H, W, C, ld, PH, PW, dil,
             PH * PW, dtype_code(a), stream_ptr())
        return ga, gb, None, None, None


def correlation(in1, in2, patch_h, patch_w, dilation_patch=1):
    return _CorrFn.apply(in1, in2, patch_h, patch_w, dilation_patch)


# ============================================================================ convolution
import weakref

class StepContext:
    """Per-training-step services installed by train.TrainStep (all optional; ops work without one):
      * zero arena: every small zero-initialised workspace of the step (statistics replicas, packed f32 weight-gradient
        accumulators, dscale/dshift replicas) is bump-allocated from ONE buffer that a single memset clears per step;
      * frozen weight packs: all weights are packed by one batched launch at step start;
      * direct gradients: parameter gradients are written (accumulated) straight into the flat gradient buffer;
      * deferred num_batches_tracked increments (one multi-tensor add per step).
    """

    def __init__(self, device):
        self.device = device
        self.arena = None
        self.offset = 0
        self.measured = 0
        self.frozen_pack = False
        self.direct_grads = False
        self.nbt = None          # list of (tensor, increment) recorded while measuring
        self.side = None         # second HIP stream: weight gradients run beside the data-gradient chain
        self.keep = []           # tensors the side stream still reads (kept alive until join())
        self.unpacks = []        # deferred weight-gradient unpack descriptors of the running step (direct_grads)
        self.unpack_key = None   # the descriptor rows the cached device table was built from
        self.unpack_desc = None
        self.seed = None         # device-resident dropout seed of the owning TrainStep (rng_seed_tensor)
        # Deferred weight gradients: nothing reads a weight gradient before the optimizer, so every layer's launch is queued
        # (with its operands kept alive) and join() issues them grouped by kernel instantiation — one grid per bucket instead
        # of ~200 per-layer launches that fill a quarter of the chip each (sdhip_conv2d_wgrad_group)
        self.defer_wgrad = not _lib.DIAG_NO_WGRAD_GROUP
        self.wq = []             # [(WgradItem fields...)] of the running step
        self.wq_keep = []        # operands of the queued launches
        self.wq_dt = None
        # single GPU: the queue is flushed onto the side stream, with grids limited to part of the chip, when the backward pass
        # reaches the DenseNet towers (overlap_point) — their ~400 small dependent kernels leave most CUs idle
        self.overlap = False

    def flush_wgrads(self, max_wg=0):
        """Issue the queued weight-gradient launches (grouped) on the current stream; max_wg > 0 limits every grid."""
        if not self.wq:
            return
        import ctypes
        items = (_lib.WgradItem * len(self.wq))()
        for it, row in zip(items, self.wq):
            (it.x, it.dy, it.dw_packed, it.dbias, it.in_scale, it.in_shift, it.B, it.H, it.W, it.Cin, it.ldx, it.Ho, it.Wo, it.Cout,
             it.lddy, it.kh, it.kw, it.stride, it.dil, it.pad_t, it.pad_l, it.D, it.Do, it.kd, it.sd, it.pad_d, it.in_relu, it.groups) = row
        n, dt = len(self.wq), self.wq_dt
        self.wq = []
        try:
            call("sdhip_conv2d_wgrad_group", ctypes.cast(items, ctypes.c_void_p), n, int(max_wg), dt, stream_ptr())
        finally:
            self.wq_keep.clear()

    def flush_to_side(self, min_items=12, max_wg=0):
        """Data parallel: the main stream spends most of the backward pass in latency-bound sync-BN exchanges — buckets of
        queued weight gradients run beside it on the side stream as soon as they are worth a grid."""
        if self.side is None or len(self.wq) < min_items:
            return
        main = torch.cuda.current_stream()
        self.side.wait_stream(main)                  # the queued operands were produced on the main stream
        self.keep.extend(self.wq_keep)               # the caching allocator must not recycle them before join()
        with torch.cuda.stream(self.side):
            self.flush_wgrads(max_wg)

    def overlap_point(self):
        """Called where the backward pass enters a long chain of small dependent kernels (a DenseNet block): what is queued
        so far goes to the side stream in grids of at most TUNE_OVERLAP_WG workgroups, which leave the other CUs to the chain."""
        if self.overlap:
            self.flush_to_side(min_items=1, max_wg=_lib.TUNE_OVERLAP_WG)

    def join(self):
        """Call after backward, before the optimizer: the queued weight gradients are launched (grouped), the main stream
        waits for the side stream, then ONE launch adds every packed weight-gradient accumulator of the step into the flat
        gradient buffer."""
        self.flush_wgrads()
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        if self.unpacks:
            key = tuple(self.unpacks)
            if key != self.unpack_key:   # arena offsets and gradient slices repeat step after step: built once
                if torch.cuda.is_current_stream_capturing():
                    raise _lib.SdhipError("weight-gradient layout changed between the warm-up step and graph capture")
                self.unpack_desc = torch.tensor(self.unpacks, dtype=torch.int64, device=self.device)
                self.unpack_key = key
            call("sdhip_conv_unpack_batch", ptr(self.unpack_desc), len(self.unpacks), self.unpack_dt, stream_ptr())
            self.unpacks = []
        self.keep.clear()

    def begin_step(self):
        self.unpacks = []
        self.wq = []
        self.wq_keep.clear()
        if self.arena is not None:
            self.offset = 0
            self.arena.zero_()

    def zeros(self, shape, dtype):
        n = 1
        for d in shape:
            n *= d
        nbytes = ((n * torch.empty((), dtype=dtype).element_size() + 255) // 256) * 256
        if self.arena is None:
            self.measured += nbytes
            return torch.zeros(shape, dtype=dtype, device=self.device), False
        if self.offset + nbytes > self.arena.numel():
            raise _lib.SdhipError("zero arena exhausted (%d + %d > %d)" % (self.offset, nbytes, self.arena.numel()))
        t = self.arena[self.offset:self.offset + n * torch.empty((), dtype=dtype).element_size()].view(dtype).view(shape)
        self.offset += nbytes
        return t, True

    def allocate_arena(self):
        self.arena = torch.zeros(int(self.measured * 1.05) + 4096, dtype=torch.uint8, device=self.device)


_ctx = [None]


def set_step_context(ctx):
    _ctx[0] = ctx


def overlap_point():
    c = _ctx[0]
    if c is not None:
        c.overlap_point()


def _zeros(shape, dtype, device):
    """(tensor of zeros, prezeroed flag for the C ABI)."""
    c = _ctx[0]
    if c is not None:
        return c.zeros(shape, dtype)
    return torch.zeros(shape, dtype=dtype, device=device), False


def _grad_target(param):
    """The flat-gradient slice to accumulate a parameter gradient into, or None (then the Function returns the gradient)."""
    c = _ctx[0]
    if c is not None and c.direct_grads and param is not None and param.is_leaf and param.grad is not None:
        return param.grad
    return None


_pack_cache = {}   # id(weight parameter) -> (weakref, {(mode, dtype): (version, packed)})
_pack_generation = [0]


def invalidate_packed_weights():
    """Call after parameters were changed outside autograd's version tracking (the fused Adam kernel)."""
    _pack_generation[0] += 1


def _cache_entry(weight):
    k = id(weight)
    hit = _pack_cache.get(k)
    if hit is None or hit[0]() is not weight:
        hit = (weakref.ref(weight, lambda _r, k=k: _pack_cache.pop(k, None)), {})
        _pack_cache[k] = hit
    return hit[1]


def _pack_params(kind, mode, Cout, Cin, T):
    """(M, K, stride_m, stride_k, flip) of include/sdhip.h for a Conv2d / stride-1 ConvTranspose2d weight."""
    if kind == 'conv':       # weight (Cout, Cin, kh, kw)
        return (Cout, Cin, Cin * T, T, 0) if mode == 'fwd' else (Cin, Cout, T, Cin * T, 1)
    if kind == 'deconv':     # weight (Cin, Cout, kh, kw), run as a correlation with flipped taps
        return (Cout, Cin, T, Cout * T, 1) if mode == 'fwd' else (Cin, Cout, Cout * T, T, 0)
    raise _lib.SdhipError("unknown conv kind %r" % kind)


def packed_weight(weight, kind, mode, dtype):
    """Pack an f32 parameter for the kernels (cached per parameter version, repacked after each optimizer step)."""
    Cout, Cin = (weight.shape[0], weight.shape[1]) if kind == 'conv' else (weight.shape[1], weight.shape[0])
    kd = weight.shape[2] if weight.dim() == 5 else 1
    T = weight.shape[-2] * weight.shape[-1]
    ent = _cache_entry(weight)
    key = (kind, mode, dtype)
    hit = ent.get(key)
    c = _ctx[0]
    if hit is not None and c is not None and c.frozen_pack:
        return hit[1]           # packed by the batched launch at step start (train.TrainStep.pack_all)
    ver = (weight._version, _pack_generation[0])
    if hit is not None and hit[0] == ver and not torch.cuda.is_current_stream_capturing():
        return hit[1]
    dt = _lib.BF16 if dtype == torch.bfloat16 else _lib.F32
    w = weight.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    rows = _pack_rows(w, kind, mode, dt)
    per = _lib.packed_elems(rows[0][2], rows[0][3], T, dt)
    buf = torch.empty(per * kd, dtype=dtype, device=weight.device)
    es = buf.element_size()
    for src_off, dst_blk, M, K, T_, sm, sk, flip in rows:   # one launch per depth tap (3-D) / one in all (2-D)
        call("sdhip_conv_pack_weights", ctypes_ptr(w.data_ptr() + 4 * src_off), ctypes_ptr(buf.data_ptr() + es * per * dst_blk),
             M, K, T_, sm, sk, flip, dt, stream_ptr())
    ent[key] = (ver, buf)
    return buf


def ctypes_ptr(addr):
    import ctypes
    return ctypes.c_void_p(addr)


def _pack_rows(w, kind, mode, dt):
    """[(src element offset, destination depth block, M, K, T, stride_m, stride_k, flip)] — one row per depth tap.
    A 3-D weight (.., kd, kh, kw) is packed [kd][q][t][m][c]; with flip (data-grad / transposed orientation) the depth
    taps are reversed as well as the in-plane taps."""
    Cout, Cin = (w.shape[0], w.shape[1]) if kind == 'conv' else (w.shape[1], w.shape[0])
    kd = w.shape[2] if w.dim() == 5 else 1
    T = w.shape[-2] * w.shape[-1]
    M, K, sm, sk, flip = _pack_params(kind, mode, Cout, Cin, T * kd)
    # _pack_params' strides are in units of "taps per (m,k) pair" = kd*T for volumes; the tap index inside one depth tap is < T
    rows = []
    for kdi in range(kd):
        rows.append((kdi * T, (kd - 1 - kdi) if flip else kdi, M, K, T, sm, sk, flip))
    return rows


def pack_descriptors(dtype, owners=None):
    """int64 [n][8] descriptor table {src, dst, M, K, T, stride_m, stride_k, flip} of the cached packs of `dtype` that belong
    to the parameters in `owners` (an iterable of tensors; None: every cached pack).  A training step must pass ITS model's
    parameters: the table is replayed step after step (inside a hipGraph), so a row of another model's weight — or of a
    per-step temporary such as the space-to-depth stem weight — would keep reading and writing that tensor's memory after
    it has been freed (a GPU memory fault once the allocator has returned the block: found by the full test suite, where
    several models are alive at once)."""
    own = None if owners is None else {id(t) for t in owners}
    rows = []
    # (a snapshot: the weak-reference callbacks of parameters that die meanwhile — a discarded model being collected — pop
    #  their entries, which must not happen under a live dictionary iterator)
    for ref, ent in list(_pack_cache.values()):
        w = ref()
        if w is None or (own is not None and id(w) not in own):
            continue
        for (kind, mode, dt), (_, buf) in list(ent.items()):
            if dt != dtype:
                continue
            dtc = _lib.BF16 if dt == torch.bfloat16 else _lib.F32
            if kind == 'phase':     # the eight sub-pixel sub-kernels of a 3x3x3 stride-2 weight: one single-tap row per tap
                for src_off, dst_off, M, K, T, sm, sk, flip in _phase_rows(w, dtc)[0]:
                    rows.append([w.data_ptr() + 4 * src_off, buf.data_ptr() + buf.element_size() * dst_off, M, K, T, sm, sk, flip])
                continue
            prow = _pack_rows(w, kind, mode, dtc)
            per = _lib.packed_elems(prow[0][2], prow[0][3], prow[0][4], dtc)
            for src_off, dst_blk, M, K, T, sm, sk, flip in prow:
                rows.append([w.data_ptr() + 4 * src_off, buf.data_ptr() + buf.element_size() * per * dst_blk, M, K, T, sm, sk, flip])
    return rows


class ConvSpec:
    """Geometry of one convolution in correlation form (what the kernels take)."""
    __slots__ = ("kind", "kh", "kw", "stride", "dil", "pad_t", "pad_l", "Ho", "Wo", "D", "Do", "kd", "sd", "pad_d")

    def __init__(self, kind, kh, kw, stride, dil, pad_t, pad_l, Ho, Wo, D=1, Do=1, kd=1, sd=1, pad_d=0):
        self.kind, self.kh, self.kw, self.stride, self.dil = kind, kh, kw, stride, dil
        self.pad_t, self.pad_l, self.Ho, self.Wo = pad_t, pad_l, Ho, Wo
        # depth axis of 3-D convolutions over [B][D][H][W][C] volumes (tensors of B*D "images"); 2-D: all ones / zero
        self.D, self.Do, self.kd, self.sd, self.pad_d = D, Do, kd, sd, pad_d

    def depth(self):
        return (self.D, self.Do, self.kd, self.sd, self.pad_d)


def _conv_launch(x, ldx, wp, y, ldy, bias, in_scale, in_shift, stats, B, H, W, Cin, Ho, Wo, Cout,
                 kh, kw, stride, dil, pad_t, pad_l, in_relu, groups, act, accumulate, nrep=1, depth=(1, 1, 1, 1, 0)):
    """stats: None, or f64 [nrep][groups][2][>=Cout] (or a [groups][2][C] slice of a slab with nrep=1).
    B is the true batch; depth = (D, Do, kd, sd, pad_d) for volumes."""
    sld = stats.stride(-2) if stats is not None else 0
    call("sdhip_conv2d_fwd", ptr(x), ptr(wp), ptr(y), ptr(bias), ptr(in_scale), ptr(in_shift), ptr(stats), sld, nrep,
         B, H, W, Cin, ldx, Ho, Wo, Cout, ldy, kh, kw, stride, dil, pad_t, pad_l, *depth,
         int(in_relu), groups, act, int(accumulate), dtype_code(x), stream_ptr())


NREP = _lib.NREP

from . import parallel  # noqa: E402


def _sync_stats(stats, nrep):
    """Data parallel: all-reduce the f64 (sum, sum of squares) over ranks (sync-BN) — in place on the replica buffer when it
    is dense (the replicas are folded by whoever consumes the buffer, as on one GPU); a strided view is folded into a
    compact copy first."""
    if stats is None or parallel.world_size() == 1:
        return stats, nrep
    if stats.is_contiguous():
        parallel.all_reduce_sum_(stats)
        return stats, nrep
    G, _, C = stats.shape[-3:]
    compact, pz = _zeros((G, 2, C), torch.float64, stats.device)   # zero arena of the step: no fill launch per layer
    if not pz:
        compact.zero_()
    call("sdhip_stats_replica_sum", ptr(stats), ptr(compact), nrep, G, C, stats.stride(-2), C, stream_ptr())
    parallel.all_reduce_sum_(compact)
    return compact, 1


def _bn_finalize(stats, nrep, bn, count, groups, synced=False):
    """(scale, shift, mean, invstd), each f32 [groups][C]; stats None => eval mode (running statistics).
    `count` is the LOCAL element count per group; under data parallelism statistics and count become global."""
    C = bn.num_features
    dev = bn.weight.device
    if not synced:      # `synced`: the statistics were already summed over ranks (slab statistics of a dense block)
        stats, nrep = _sync_stats(stats, nrep)
    count = parallel.global_count(count)
    out = [torch.empty((groups, C), dtype=torch.float32, device=dev) for _ in range(4)]
    if stats is not None:
        mom = 0.1 if bn.momentum is None else bn.momentum
        if bn.num_batches_tracked is not None:
            c = _ctx[0]
            if c is not None and c.nbt is not None and c.arena is not None:
                pass                                                  # applied by one multi-tensor add per step (train.TrainStep)
            else:
                if c is not None and c.nbt is not None:
                    c.nbt.append((bn.num_batches_tracked, groups))   # recorded during the measuring step
                bn.num_batches_tracked += groups
        call("sdhip_bn_finalize", ptr(stats), stats.stride(-2), nrep, ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean),
             ptr(bn.running_var), ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]), C, groups, float(count),
             float(bn.eps), float(mom), stream_ptr())
    else:
        call("sdhip_bn_finalize", None, 0, 1, ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var),
             ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]), C, groups, float(count), float(bn.eps), 0.0, stream_ptr())
    return out


def _fused_bn():
    """Consumer-side finalize kernels.  They also serve under data parallelism: the sync-BN exchange all-reduces the
    producer's replica buffer in place, between the launch that fills it and the launch that consumes it."""
    return not _lib.DIAG_NO_FUSED_BN


def _bn_track(bn, groups):
    """BatchNorm.num_batches_tracked bookkeeping of one train-mode normalisation (see _bn_finalize)."""
    if bn.num_batches_tracked is None:
        return
    c = _ctx[0]
    if c is not None and c.nbt is not None and c.arena is not None:
        return                                                    # applied by one multi-tensor add per step (train.TrainStep)
    if c is not None and c.nbt is not None:
        c.nbt.append((bn.num_batches_tracked, groups))           # recorded during the measuring step
    bn.num_batches_tracked += groups


def bn_backward_two_phase(gy, ldg, x, ldx, gx, ldgx, scale, shift, mean, invstd, gamma, beta, npix, C, groups, act, count, dt):
    """Train-mode backward of y = act(BatchNorm(x)): (1) reductions only, [data parallel: the replica sums are all-reduced
    in place], (2) ONE pass that derives the statistics gradient from the replica sums itself and writes the complete
    gradient of x (+ dgamma / dbeta).  `count`: the global element count per group.
    Returns (dgamma, dbeta), each None when accumulated straight into the flat gradient buffer."""
    dev = scale.device
    both, pz = _zeros((2, NREP, groups, C), torch.float32, dev)
    dsc, dsh = both[0], both[1]
    call("sdhip_affine_act_bwd", ptr(gy), ldg, ptr(x), ldx, None, 0, ptr(scale), ptr(shift), ptr(dsc), ptr(dsh), NREP,
         npix, C, groups, act, 0, int(pz), dt, stream_ptr())
    parallel.all_reduce_sum_(both)            # sync-BN: the replica sums become global in place (`count` is the global count)
    tg, tb = _grad_target(gamma), _grad_target(beta)
    direct = tg is not None and tb is not None
    dgamma = tg if direct else torch.empty(C, dtype=torch.float32, device=dev)
    dbeta = tb if direct else torch.empty(C, dtype=torch.float32, device=dev)
    call("sdhip_bn_bwd_apply_fin", ptr(gy), ldg, ptr(x), ldx, ptr(gx), ldgx, ptr(scale), ptr(shift), ptr(dsc), ptr(dsh), NREP,
         ptr(gamma), ptr(mean), ptr(invstd), ptr(dgamma), ptr(dbeta), int(direct), parallel.param_scale(), npix, C, groups,
         float(count), act, dt, stream_ptr())
    return (None, None) if direct else (dgamma, dbeta)


def _bn_backward(gy, ldg, x, ldx, gx, ldgx, scale, shift, mean, invstd, gamma, npix, C, groups, act, count, train,
                 dt, accumulate_gx=False, dstats=None, accumulate_dstats=False, beta=None):
    """Backward of y = act(x*scale + shift) with scale/shift from batch statistics.
    Writes gx (+)= gy*act'*scale and returns (dgamma, dbeta, dstats[groups][2][C]); the caller still has to add the
    statistics path dstats[0] + 2*x*dstats[1] (sdhip_stats_fix) to the gradient of whatever produced x."""
    dev = scale.device
    both, pz = _zeros((2, NREP, groups, C), torch.float32, dev)
    dsc, dsh = both[0], both[1]
    call("sdhip_affine_act_bwd", ptr(gy), ldg, ptr(x), ldx, ptr(gx), ldgx, ptr(scale), ptr(shift), ptr(dsc), ptr(dsh), NREP,
         npix, C, groups, act, int(accumulate_gx), int(pz), dt, stream_ptr())
    tg, tb = _grad_target(gamma), _grad_target(beta)
    direct = tg is not None and tb is not None
    dgamma = tg if direct else torch.empty(C, dtype=torch.float32, device=dev)
    dbeta = tb if direct else torch.empty(C, dtype=torch.float32, device=dev)
    par_flag = 2 if direct else 0
    if dstats is None:
        dstats = torch.empty((groups, 2, C), dtype=torch.float64, device=dev)
        accumulate_dstats = False
    if parallel.world_size() > 1 and train:
        # sync-BN backward: dgamma/dbeta from the LOCAL sums, the statistics gradient from the GLOBAL sums
        call("sdhip_bn_finalize_bwd", ptr(dsc), ptr(dsh), NREP, ptr(gamma), ptr(mean), ptr(invstd), ptr(dgamma), ptr(dbeta),
             None, C, par_flag, C, groups, float(count), 1, stream_ptr())
        parallel.all_reduce_sum_(both)                         # [2][NREP][groups][C] in place: the finalize folds the replicas
        call("sdhip_bn_finalize_bwd", ptr(dsc), ptr(dsh), NREP, ptr(gamma), ptr(mean), ptr(invstd), None, None,
             ptr(dstats), dstats.stride(-2), int(accumulate_dstats), C, groups, float(parallel.global_count(count)), 1,
             stream_ptr())
        return (None, None, dstats) if direct else (dgamma, dbeta, dstats)
    call("sdhip_bn_finalize_bwd", ptr(dsc), ptr(dsh), NREP, ptr(gamma), ptr(mean), ptr(invstd), ptr(dgamma), ptr(dbeta),
         ptr(dstats), dstats.stride(-2), int(accumulate_dstats) | par_flag, C, groups, float(count), int(train), stream_ptr())
    return (None, None, dstats) if direct else (dgamma, dbeta, dstats)


class GradSlot:
    """Carries one gradient contribution from the autograd node that produces it FIRST in the backward pass to the node
    whose data-gradient launch can add to it in place.  Conv2DownUp (models/dsnet_t2.py:80-117): x1 feeds c2 and, as a skip,
    d4 — d4's backward runs before c2's (it was created later), parks the skip gradient here instead of handing it to
    autograd, and c2's data gradient is launched with accumulate = 1 on that buffer: the sum of the two contributions
    costs no pass of its own (autograd would run an elementwise add over the full-resolution map)."""
    __slots__ = ("g", "exclusive")

    def __init__(self, exclusive):
        # exclusive: the parked tensor is referenced by nobody else (it is the fresh data gradient of the node behind the
        # parking one), so the sum may be formed IN it; otherwise it is only read (sdhip_conv2d_fwd_add) — an incoming
        # gradient from outside the block may be shared with another branch of the graph
        self.g = None
        self.exclusive = exclusive


BN_SLOT_MAX_BYTES = _lib.TUNE_BN_SLOT_MAX_MB << 20


class BNSlot:
    """Links a convbn + ReLU node to the ONE node that consumes its output, so that the consumer's data-gradient launch —
    which holds every value of dL/d relu(bn(u)) once — also takes the two reductions the BatchNorm backward starts with
    (sdhip_conv2d_fwd_bnbwd) and the producer's backward skips its reduction pass over that gradient and u.
    The producer fills (u, scale, shift) in its forward; the consumer's backward fills (sums, gptr); the producer uses the
    sums only if the gradient it is handed IS the tensor the consumer wrote (gptr) — any other gradient route (an extra
    consumer, autograd summing contributions) silently falls back to the two-pass form."""
    __slots__ = ("u", "ldu", "scale", "shift", "groups", "C", "sums", "gptr")

    def __init__(self):
        self.u = self.scale = self.shift = self.sums = None
        self.ldu = self.groups = self.C = 0
        self.gptr = 0


def _conv_backward(ctx_spec, xv, ldx, weight, g, ldg, in_scale, in_shift, in_relu, groups, need_x, need_w, bias=None, acc_into=None,
                   addend=None, bn_slot=None):
    has_bias = bias is not None
    """dgrad (w.r.t. the post-prologue input) and wgrad of one conv; returns (g_post, gw, gb)."""
    spec = ctx_spec
    Bimg, Cin, H, W = xv.shape
    B = Bimg // spec.D
    Cout = weight.shape[0] if spec.kind == 'conv' else weight.shape[1]
    dt = dtype_code(xv)
    gpost = gw = gb = None
    if need_x:
        wd = packed_weight(weight, spec.kind, 'dgrad', xv.dtype)
        fused_bn = False
        # Measured (profiles/r02_step_summary.txt): the epilogue's reads of u sit exposed behind the MFMA loop, so on maps
        # of more than ~40 MB (and in the one-workgroup-per-CU 5x5 kernel at any size) it costs more than the streaming
        # pass it replaces (full resolution, 32 channels: +33 us against a 27 us pass; 5x5 / 64 channels: +126 against 60);
        # below that the pass is launch- and latency-bound and the fusion wins (64 channels at 128x256: +3.5 against 18)
        if (bn_slot is not None and bn_slot.u is not None and bn_slot.C == Cin and xv.dtype == torch.bfloat16 and spec.stride == 1
                and spec.sd == 1 and spec.kd == 1 and spec.D == 1 and tuple(bn_slot.u.shape) == (Bimg, Cin, H, W)
                and spec.kh * spec.kw <= 9 and Bimg * H * W * Cin * 2 <= BN_SLOT_MAX_BYTES):
            # the input of this convolution was relu(bn(u)): the launch that writes its gradient also takes that
            # BatchNorm's two backward reductions (and adds a parked skip gradient on the way)
            other = acc_into if acc_into is not None else addend
            gpost, ldgp = alloc_nhwc(Bimg, Cin, H, W, xv.dtype, xv.device)
            sums = _zeros((NREP, bn_slot.groups, 2, Cin), torch.float64, xv.device)[0]
            ov, ldo = nhwc_view(other) if other is not None else (None, 0)
            rc = _lib._lib.sdhip_conv2d_fwd_bnbwd(ptr(g), ptr(wd), ptr(gpost), ptr(sums), Cin, NREP, ptr(bn_slot.u), bn_slot.ldu,
                                                  ptr(bn_slot.scale), ptr(bn_slot.shift), ptr(ov), ldo, B, spec.Ho, spec.Wo, Cout, ldg,
                                                  H, W, Cin, ldgp, spec.kh, spec.kw, spec.dil, spec.dil * (spec.kh - 1) - spec.pad_t,
                                                  spec.dil * (spec.kw - 1) - spec.pad_l, bn_slot.groups, 0, dtype_code(xv), stream_ptr())
            if rc == 0:
                bn_slot.sums, bn_slot.gptr = sums, gpost.data_ptr()
                fused_bn = True
            elif rc != _lib.ERR_UNSUPPORTED:
                raise _lib.SdhipError("sdhip_conv2d_fwd_bnbwd failed (%d): %s" % (rc, _lib._lib.sdhip_last_error().decode()))
        # Conv3d(k=3, stride 2, padding 1) over even extents (hourglass conv1 / conv3, stackhourglass.py:13-19): its data gradient is
        # the transposed convolution that doubles every extent — eight sub-pixel phases of 1..8 taps over dY itself (27 taps per dY
        # voxel), not a stride-1 correlation over the zero-stuffed dY (8 x 27: seven of eight products are zeros)
        if (not fused_bn and acc_into is None and addend is None and spec.kind == 'conv' and spec.stride == 2 and spec.sd == 2
                and spec.kd == 3 and spec.kh == 3 and spec.kw == 3 and spec.dil == 1 and spec.pad_t == 1 and spec.pad_l == 1
                and spec.pad_d == 1 and spec.D == 2 * spec.Do and H == 2 * spec.Ho and W == 2 * spec.Wo and not _lib.DIAG_NO_PHASE_DGRAD):
            gpost, ldgp = alloc_nhwc(Bimg, Cin, H, W, xv.dtype, xv.device)
            packs, _keep = _phase_packs(weight, xv.dtype)       # (Cout, Cin, 3,3,3) IS the transposed convolution's (in, out, ...) layout
            for (pd, ph, pw), wpk in zip(_PHASES, packs):
                call("sdhip_conv2d_fwd_phase", ptr(g), ptr(wpk), ptr(gpost), None, 0, 1, B, spec.Ho, spec.Wo, Cout, ldg, Cin, ldgp,
                     1 + ph, 1 + pw, spec.Do, 1 + pd, 1, pd, ph, pw, dt, stream_ptr())
            if need_w:
                gw, gb = wgrad(xv, ldx, g, ldg, weight, bias, spec, in_scale, in_shift, in_relu, groups)
            return gpost, gw, gb
        if fused_bn:
            pass
        elif acc_into is not None:
            gpost, ldgp = nhwc_view(acc_into)     # holds the other contribution: this launch adds to it
        else:
            gpost, ldgp = alloc_nhwc(Bimg, Cin, H, W, xv.dtype, xv.device)
        pt = spec.dil * (spec.kh - 1) - spec.pad_t
        pl = spec.dil * (spec.kw - 1) - spec.pad_l
        pd = (spec.kd - 1) - spec.pad_d
        gsrc, ldsrc, Dg, Hg, Wg = g, ldg, spec.Do, spec.Ho, spec.Wo
        if spec.stride != 1 or spec.sd != 1:
            # strided convolution: its data gradient is a stride-1 correlation (flipped weights) over the zero-stuffed dY
            Dg, Hg, Wg = (spec.Do - 1) * spec.sd + 1, (spec.Ho - 1) * spec.stride + 1, (spec.Wo - 1) * spec.stride + 1
            gsrc = empty_nhwc(B * Dg, Cout, Hg, Wg, xv.dtype, xv.device)
            call("sdhip_stuff", ptr(g), ldg, ptr(gsrc), Cout, B, spec.Do, spec.Ho, spec.Wo, Cout, spec.sd, spec.stride, 1, dt,
                 stream_ptr())
            ldsrc = Cout
        fused_add = fused_bn
        if not fused_bn and addend is not None and spec.kd == 1 and spec.D == 1 and spec.dil == 1 and gsrc is g:
            av, lda = nhwc_view(addend)
            rc = _lib._lib.sdhip_conv2d_fwd_add(ptr(gsrc), ptr(wd), ptr(gpost), ptr(av), lda, B, Hg, Wg, Cout, ldsrc, H, W, Cin, ldgp,
                                                spec.kh, spec.kw, pt, pl, dtype_code(xv), stream_ptr())
            if rc not in (0, _lib.ERR_UNSUPPORTED):
                raise _lib.SdhipError("sdhip_conv2d_fwd_add failed (%d): %s" % (rc, _lib._lib.sdhip_last_error().decode()))
            fused_add = rc == 0
        if not fused_add:
            _conv_launch(gsrc, ldsrc, wd, gpost, ldgp, None, None, None, None, B, Hg, Wg, Cout, H, W, Cin,
                         spec.kh, spec.kw, 1, spec.dil, pt, pl, False, 1, 0, acc_into is not None, 1, (Dg, spec.D, spec.kd, 1, pd))
            if addend is not None:            # no kernel on this shape's path adds a second tensor: one elementwise pass
                gpost = add(gpost, addend)

    if need_w:
        gw, gb = wgrad(xv, ldx, g, ldg, weight, bias, spec, in_scale, in_shift, in_relu, groups)
    return gpost, gw, gb


def wgrad(xv, ldx, g, ldg, weight, bias, spec, in_scale=None, in_shift=None, in_relu=False, groups=1):
    """Weight (and bias) gradient of one conv.  Returns (gw, gb), each None when it was accumulated straight into the
    flat gradient buffer (StepContext.direct_grads).  With a StepContext side stream the launches go there: nothing
    downstream of a weight gradient runs before the optimizer, so it overlaps the latency-bound data-gradient chain."""
    c = _ctx[0]
    if c is not None and c.side is not None and c.direct_grads and not c.defer_wgrad:
        main = torch.cuda.current_stream()
        c.side.wait_stream(main)                     # g and x are produced on the main stream
        c.keep.append((xv, g, in_scale, in_shift))   # the caching allocator must not recycle them before join()
        with torch.cuda.stream(c.side):
            return _wgrad_impl(xv, ldx, g, ldg, weight, bias, spec, in_scale, in_shift, in_relu, groups)
    return _wgrad_impl(xv, ldx, g, ldg, weight, bias, spec, in_scale, in_shift, in_relu, groups)


def _wgrad_impl(xv, ldx, g, ldg, weight, bias, spec, in_scale, in_shift, in_relu, groups):
    Bimg, Cin, H, W = xv.shape
    B = Bimg // spec.D
    Cout = weight.shape[0] if spec.kind == 'conv' else weight.shape[1]
    dt = dtype_code(xv)
    T = spec.kh * spec.kw
    per = _lib.packed_elems(Cout, Cin, T, dt)
    acc, pz = _zeros((per * spec.kd,), torch.float32, xv.device)
    tw = _grad_target(weight)
    tb = _grad_target(bias) if bias is not None else None
    gb = None
    if bias is not None:
        if tb is not None:
            dbias, dbias_pz = tb, True       # the flat gradient buffer is zeroed once per step; the kernel adds
        else:
            dbias, dbias_pz = _zeros((Cout,), torch.float32, xv.device)
            gb = dbias
    else:
        dbias, dbias_pz = None, True
    if bias is not None and pz != dbias_pz:   # one flag covers both buffers: zero the not-yet-zeroed one here
        if not pz:
            acc.zero_()
        elif tb is None:
            dbias.zero_()
        pz = True
    c = _ctx[0]
    if (c is not None and c.defer_wgrad and c.direct_grads and c.arena is not None and tw is not None and pz
            and (bias is None or tb is not None)):
        # queued: launched by StepContext.join() in one grid per kernel instantiation; the operands stay alive until then
        dpt = lambda t: t.data_ptr() if t is not None else None
        c.wq.append((dpt(xv), dpt(g), dpt(acc), dpt(dbias), dpt(in_scale), dpt(in_shift), B, H, W, Cin, ldx, spec.Ho, spec.Wo, Cout, ldg,
                     spec.kh, spec.kw, spec.stride, spec.dil, spec.pad_t, spec.pad_l) + tuple(spec.depth()) + (int(in_relu), groups))
        c.wq_keep.append((xv, g, in_scale, in_shift))
        c.wq_dt = dt
        if not c.overlap:
            c.flush_to_side()
    else:
        call("sdhip_conv2d_wgrad", ptr(xv), ptr(g), ptr(acc), ptr(dbias), ptr(in_scale), ptr(in_shift),
             B, H, W, Cin, ldx, spec.Ho, spec.Wo, Cout, ldg, spec.kh, spec.kw, spec.stride, spec.dil,
             spec.pad_t, spec.pad_l, *spec.depth(), int(in_relu), groups, int(pz), dt, stream_ptr())
    # unpack by the WEIGHT's own channel count (the activation may carry zero-padded extra channels: 8-channel images),
    # depth tap by depth tap for 3-D weights
    gw = None
    target = tw
    if target is None:
        gw = target = torch.empty_like(weight, memory_format=torch.contiguous_format)
    c = _ctx[0]
    defer = c is not None and c.direct_grads and tw is not None and pz and c.arena is not None
    for src_off, blk, M, K, T_, sm, sk, flip in _pack_rows(weight, spec.kind, 'fwd', dt):
        if defer:   # accumulators live in the step's zero arena until join(): unpack all of them in one launch there
            c.unpacks.append((acc.data_ptr() + 4 * per * blk, target.data_ptr() + 4 * src_off, M, K, T_, sm, sk, flip))
            c.unpack_dt = dt
        else:
            call("sdhip_conv_unpack_wgrad", ctypes_ptr(acc.data_ptr() + 4 * per * blk), ctypes_ptr(target.data_ptr() + 4 * src_off),
                 M, K, T_, sm, sk, flip, 1 if tw is not None else 0, dt, stream_ptr())
    return gw, gb


class _ConvFn(torch.autograd.Function):
    """y = act(conv(x) + bias)   (conv2dSame / ConvTranspose2dSame / nn.Conv2d without a BatchNorm behind it)."""

    @staticmethod
    def forward(ctx, x, weight, bias, spec, act):
        _require_gpu(x, weight)
        Bimg, Cin, H, W = x.shape
        B = Bimg // spec.D
        xv, ldx = aligned_view(x)
        Cout = weight.shape[0] if spec.kind == 'conv' else weight.shape[1]
        wp = packed_weight(weight, spec.kind, 'fwd', x.dtype)
        y, ldy = alloc_nhwc(B * spec.Do, Cout, spec.Ho, spec.Wo, x.dtype, x.device)
        _conv_launch(xv, ldx, wp, y, ldy, bias.detach() if bias is not None else None, None, None, None, B, H, W, Cin,
                     spec.Ho, spec.Wo, Cout, spec.kh, spec.kw, spec.stride, spec.dil, spec.pad_t, spec.pad_l, False, 1, act, False,
                     1, spec.depth())
        ctx.spec, ctx.act, ctx.ldx = spec, act, ldx
        ctx.save_for_backward(xv, weight, bias, y if act else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        xv, weight, bias, ysaved = ctx.saved_tensors
        spec, act = ctx.spec, ctx.act
        B = xv.shape[0] // spec.D * spec.Do      # output images
        Cout = weight.shape[0] if spec.kind == 'conv' else weight.shape[1]
        g, ldg = aligned_view(gy)
        if act:   # activation fused in the epilogue: derivative from the stored output
            g2, ldg2 = alloc_nhwc(B, Cout, spec.Ho, spec.Wo, xv.dtype, xv.device)
            call("sdhip_affine_act_bwd", ptr(g), ldg, ptr(ysaved), nhwc_view(ysaved)[1], ptr(g2), ldg2, None, None, None, None, 1,
                 B * spec.Ho * spec.Wo, Cout, 1, 1 if act == 1 else 4, 0, 0, dtype_code(xv), stream_ptr())
            g, ldg = g2, ldg2
        gx, gw, gb = _conv_backward(spec, xv, ctx.ldx, weight, g, ldg, None, None, False, 1, ctx.needs_input_grad[0],
                                    ctx.needs_input_grad[1] or (bias is not None and ctx.needs_input_grad[2]), bias)
        return gx, gw, gb, None, None


class _ConvBNActFn(torch.autograd.Function):
    """y = act(BatchNorm(conv(x))) (+ residual) as ONE autograd node: the conv epilogue sums the batch statistics,
    a per-channel kernel turns them into scale/shift (and updates the running statistics), one elementwise pass
    normalises, activates and adds the skip.  convbn / deconvbn (+ReLU, + skip add) of models/dsnet_t2.py:16-117."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, spec, bn, act, groups, in_slot=None, res_slot=None, in_bn=None, out_bn=None):
        _require_gpu(x, weight)
        ctx.in_slot, ctx.res_slot, ctx.in_bn, ctx.out_bn = in_slot, res_slot, in_bn, out_bn
        Bimg, Cin, H, W = x.shape
        Btrue = Bimg // spec.D
        B = Btrue * spec.Do                      # output images
        xv, ldx = aligned_view(x)
        Cout = weight.shape[0] if spec.kind == 'conv' else weight.shape[1]
        wp = packed_weight(weight, spec.kind, 'fwd', x.dtype)
        yraw, ldr_ = alloc_nhwc(B, Cout, spec.Ho, spec.Wo, x.dtype, x.device)
        train = bn.training
        ws = _zeros((NREP, groups, 2, Cout), torch.float64, x.device)[0] if train else None
        _conv_launch(xv, ldx, wp, yraw, ldr_, None, None, None, ws, Btrue, H, W, Cin, spec.Ho, spec.Wo, Cout,
                     spec.kh, spec.kw, spec.stride, spec.dil, spec.pad_t, spec.pad_l, False, groups, 0, False, NREP, spec.depth())
        count = (B // groups) * spec.Ho * spec.Wo
        rv, ldr = nhwc_view(residual) if residual is not None else (None, 0)
        y, ldy = alloc_nhwc(B, Cout, spec.Ho, spec.Wo, x.dtype, x.device)
        if train and _fused_bn():
            # one launch: every workgroup derives scale/shift of its channels from the statistics the conv just wrote
            scale, shift, mean, invstd = [torch.empty((groups, Cout), dtype=torch.float32, device=x.device) for _ in range(4)]
            _bn_track(bn, groups)
            parallel.all_reduce_sum_(ws)         # sync-BN: the conv epilogue's replica sums become global, in place
            call("sdhip_affine_act_bn", ptr(yraw), ldr_, ptr(y), ldy, ptr(rv), ldr, ptr(ws), ws.stride(-2), NREP, ptr(bn.weight),
                 ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var), ptr(scale), ptr(shift), ptr(mean), ptr(invstd),
                 B * spec.Ho * spec.Wo, Cout, groups, float(parallel.global_count(count)), float(bn.eps),
                 float(0.1 if bn.momentum is None else bn.momentum), act, dtype_code(x), stream_ptr())
        else:
            scale, shift, mean, invstd = _bn_finalize(ws, NREP, bn, count, groups)
            call("sdhip_affine_act", ptr(yraw), ldr_, ptr(y), ldy, ptr(rv), ldr, ptr(scale), ptr(shift), B * spec.Ho * spec.Wo,
                 Cout, groups, act, dtype_code(x), stream_ptr())
        ctx.spec, ctx.act, ctx.groups, ctx.ldx, ctx.count, ctx.train = spec, act, groups, ldx, count, train
        ctx.ldraw = ldr_
        ctx.has_res = residual is not None
        if out_bn is not None and train and act == 1 and _fused_bn() and spec.D == 1 and spec.Do == 1:
            out_bn.u, out_bn.ldu, out_bn.scale, out_bn.shift, out_bn.groups, out_bn.C = yraw, ldr_, scale, shift, groups, Cout
        ctx.save_for_backward(xv, weight, gamma, beta, yraw, scale, shift, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        xv, weight, gamma, beta, yraw, scale, shift, mean, invstd = ctx.saved_tensors
        spec, groups = ctx.spec, ctx.groups
        B = yraw.shape[0]
        Cout = yraw.shape[1]
        npix = B * spec.Ho * spec.Wo
        dt = dtype_code(xv)
        g, ldg = nhwc_view(gy)
        graw, ldgr = alloc_nhwc(B, Cout, spec.Ho, spec.Wo, xv.dtype, xv.device)
        ldraw = ctx.ldraw
        ob = ctx.out_bn
        if ob is not None and ob.sums is not None and ob.gptr == g.data_ptr() and ctx.train and _fused_bn():
            # the node that produced gy already summed it against yraw (BNSlot): straight to the second phase
            tg, tb = _grad_target(gamma), _grad_target(beta)
            direct = tg is not None and tb is not None
            dgamma = tg if direct else torch.empty(Cout, dtype=torch.float32, device=xv.device)
            dbeta = tb if direct else torch.empty(Cout, dtype=torch.float32, device=xv.device)
            parallel.all_reduce_sum_(ob.sums)
            call("sdhip_bn_bwd_apply_fin_d", ptr(g), ldg, ptr(yraw), ldraw, ptr(graw), ldgr, ptr(scale), ptr(shift), ptr(ob.sums), NREP,
                 ptr(gamma), ptr(mean), ptr(invstd), ptr(dgamma), ptr(dbeta), int(direct), parallel.param_scale(), npix, Cout, groups,
                 float(parallel.global_count(ctx.count)), ctx.act, dt, stream_ptr())
            if direct:
                dgamma = dbeta = None
            ob.sums, ob.gptr = None, 0
        elif ctx.train and ctx.act in (0, 1, 2) and _fused_bn():
            dgamma, dbeta = bn_backward_two_phase(g, ldg, yraw, ldraw, graw, ldgr, scale, shift, mean, invstd, gamma, beta, npix,
                                                  Cout, groups, ctx.act, parallel.global_count(ctx.count), dt)
        elif ctx.train and ctx.act in (0, 1, 2):
            # two-phase: reductions only (no gradient written), per-channel finalize, then ONE pass writes the complete
            # gradient of the conv output — 10 bytes per element instead of 12
            dgamma, dbeta, dS = _bn_backward(g, ldg, yraw, ldraw, None, 0, scale, shift, mean, invstd, gamma, npix, Cout,
                                             groups, ctx.act, ctx.count, True, dt, beta=beta)
            call("sdhip_bn_bwd_apply", ptr(g), ldg, ptr(yraw), ldraw, ptr(graw), ldgr, ptr(scale), ptr(shift), ptr(dS), Cout,
                 npix, Cout, groups, ctx.act, dt, stream_ptr())
        else:
            dgamma, dbeta, dS = _bn_backward(g, ldg, yraw, ldraw, graw, ldgr, scale, shift, mean, invstd, gamma, npix, Cout,
                                             groups, ctx.act, ctx.count, ctx.train, dt, beta=beta)
        acc = addend = None
        if ctx.in_slot is not None and ctx.in_slot.g is not None and ctx.needs_input_grad[0]:
            parked, ctx.in_slot.g = ctx.in_slot.g, None       # the skip consumer of x already ran: add to its contribution
            if parked.shape != xv.shape or parked.dtype != xv.dtype:
                raise _lib.SdhipError("GradSlot: parked gradient does not match the input it belongs to")
            if ctx.in_slot.exclusive:
                acc = parked
            else:
                addend = parked
        gx, gw, _ = _conv_backward(spec, xv, ctx.ldx, weight, graw, ldgr, None, None, False, 1, ctx.needs_input_grad[0],
                                   ctx.needs_input_grad[1], acc_into=acc, addend=addend, bn_slot=ctx.in_bn)
        gres = gy if ctx.has_res else None
        if gres is not None and ctx.res_slot is not None:
            ctx.res_slot.g, gres = gy, None                   # handed to the producer-side consumer of the skip tensor instead
        return gx, gw, dgamma, dbeta, gres, None, None, None, None, None, None, None, None


class _BNConvFn(torch.autograd.Function):
    """y = conv(relu(BatchNorm(x))) with the statistics of x given (DenseNet transition, models/densenet.py:119-128):
    norm + relu are the conv's fused input prologue.  Returns gradients for x and for its statistics."""

    @staticmethod
    def forward(ctx, x, stats, weight, gamma, beta, spec, bn, groups):
        _require_gpu(x, weight)
        B, Cin, H, W = x.shape
        xv, ldx = nhwc_view(x)
        Cout = weight.shape[0]
        train = bn.training
        count = (B // groups) * H * W
        scale, shift, mean, invstd = _bn_finalize(stats if train else None, 1, bn, count, groups, synced=True)
        wp = packed_weight(weight, spec.kind, 'fwd', x.dtype)
        y = empty_nhwc(B, Cout, spec.Ho, spec.Wo, x.dtype, x.device)
        _conv_launch(xv, ldx, wp, y, Cout, None, scale, shift, None, B, H, W, Cin, spec.Ho, spec.Wo, Cout,
                     spec.kh, spec.kw, spec.stride, spec.dil, spec.pad_t, spec.pad_l, True, groups, 0, False)
        ctx.spec, ctx.groups, ctx.ldx, ctx.count, ctx.train = spec, groups, ldx, count, train
        ctx.save_for_backward(xv, weight, gamma, beta, scale, shift, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        xv, weight, gamma, beta, scale, shift, mean, invstd = ctx.saved_tensors
        spec, groups = ctx.spec, ctx.groups
        B, Cin, H, W = xv.shape
        dt = dtype_code(xv)
        g, ldg = nhwc_view(gy)
        gpost, gw, _ = _conv_backward(spec, xv, ctx.ldx, weight, g, ldg, scale, shift, True, groups, True, True)
        gx = empty_nhwc(B, Cin, H, W, xv.dtype, xv.device)
        dgamma, dbeta, dS = _bn_backward(gpost, nhwc_view(gpost)[1], xv, ctx.ldx, gx, Cin, scale, shift, mean, invstd, gamma,
                                         B * H * W, Cin, groups, 1, ctx.count, ctx.train, dt, beta=beta)
        return gx, (dS if ctx.train else None), gw, dgamma, dbeta, None, None, None


class _BNActFn(torch.autograd.Function):
    """y = act(BatchNorm(x)) with the statistics of x given (norm5 of the DenseNet, models/densenet.py:239-241)."""

    @staticmethod
    def forward(ctx, x, stats, gamma, beta, bn, act, groups):
        _require_gpu(x)
        B, C, H, W = x.shape
        xv, ldx = nhwc_view(x)
        train = bn.training
        count = (B // groups) * H * W
        scale, shift, mean, invstd = _bn_finalize(stats if train else None, 1, bn, count, groups, synced=True)
        y = empty_nhwc(B, C, H, W, x.dtype, x.device)
        call("sdhip_affine_act", ptr(xv), ldx, ptr(y), C, None, 0, ptr(scale), ptr(shift), B * H * W, C, groups, act,
             dtype_code(x), stream_ptr())
        ctx.cfg = (ldx, act, groups, count, train)
        ctx.save_for_backward(xv, gamma, beta, scale, shift, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        xv, gamma, beta, scale, shift, mean, invstd = ctx.saved_tensors
        ldx, act, groups, count, train = ctx.cfg
        B, C, H, W = xv.shape
        g, ldg = nhwc_view(gy)
        gx = empty_nhwc(B, C, H, W, xv.dtype, xv.device)
        dgamma, dbeta, dS = _bn_backward(g, ldg, xv, ldx, gx, C, scale, shift, mean, invstd, gamma, B * H * W, C, groups, act,
                                         count, train, dtype_code(xv), beta=beta)
        return gx, (dS if train else None), dgamma, dbeta, None, None, None


def conv1x1_cat_forward(segs, weight, bias, act):
    """act(conv1x1(cat([upsample_nearest(x_i, 2^us_i)], dim=1)) + bias) without building the concatenation or the
    upsampled maps (sdhip_conv1x1_cat_fwd).  segs: [(tensor (B, c_i, H >> us_i, W >> us_i), us_i)], at most two; bf16."""
    (x0, us0) = segs[0]
    x1, us1 = segs[1] if len(segs) > 1 else (None, 0)
    B, c0 = x0.shape[0], x0.shape[1]
    H, W = x0.shape[2] << us0, x0.shape[3] << us0
    v0, ld0 = aligned_view(x0)
    v1, ld1 = aligned_view(x1) if x1 is not None else (None, 0)
    c1 = x1.shape[1] if x1 is not None else 0
    Cout = weight.shape[0]
    wp = packed_weight(weight, 'conv', 'fwd', x0.dtype)
    y, ldy = alloc_nhwc(B, Cout, H, W, x0.dtype, x0.device)
    call("sdhip_conv1x1_cat_fwd", ptr(v0), ld0, c0, us0, ptr(v1), ld1, c1, us1, ptr(wp), ptr(y), ldy,
         ptr(bias.detach()) if bias is not None else None, B, H, W, Cout, act, dtype_code(x0), stream_ptr())
    return y


class _UpCatConv1x1Fn(torch.autograd.Function):
    """y = act(conv1x1(cat([nearest_upsample(xs, 2^us), xf], dim=1))) — `conv1d_2(torch.cat((F.interpolate(y, scale_factor=8),
    xleft2), 1))` and its siblings (models/dsnet_t2.py:1211-1216,1262-1291,927-933) as one kernel: neither the upsampled
    map nor the concatenation is ever written.  The backward pass never touches the upsampled form either:
      grad(xs)      = dgrad1x1(sum-pool_{2^us}(g), W)[:, :c0]        (pooling and the 1x1 contraction commute)
      grad(xf)      = conv1x1(g, W[:, c0:]^T)
      dW[:, :c0]    = wgrad(xs, sum-pool(g)),   dW[:, c0:] = wgrad(xf, g)."""

    @staticmethod
    def forward(ctx, xs, xf, weight, us, act):
        _require_gpu(xs, xf, weight)
        y = conv1x1_cat_forward([(xs, us), (xf, 0)], weight, None, act)
        ctx.save_for_backward(xs, xf, weight, y if act else None)
        ctx.cfg = (us, act)
        return y

    @staticmethod
    def backward(ctx, gy):
        xs, xf, weight, ysaved = ctx.saved_tensors
        us, act = ctx.cfg
        B, Cout, H, W = gy.shape
        c0, c1 = xs.shape[1], xf.shape[1]
        h, w = xs.shape[2], xs.shape[3]
        dt = dtype_code(gy)
        g, ldg = aligned_view(gy)
        if act:
            g2, ldg2 = alloc_nhwc(B, Cout, H, W, gy.dtype, gy.device)
            call("sdhip_affine_act_bwd", ptr(g), ldg, ptr(ysaved), nhwc_view(ysaved)[1], ptr(g2), ldg2, None, None, None, None, 1,
                 B * H * W, Cout, 1, 1 if act == 1 else 4, 0, 0, dt, stream_ptr())
            g, ldg = g2, ldg2
        k = 1 << us
        gp = avgpool(g, k)                                   # mean over the 2^us x 2^us block; the sum is k*k times that
        gpv, ldgp = aligned_view(gp)
        # gradient of the small map
        wd = packed_weight(weight, 'conv', 'dgrad', gy.dtype)
        gsm, ldgs = alloc_nhwc(B, c0 + c1, h, w, gy.dtype, gy.device)
        _conv_launch(gpv, ldgp, wd, gsm, ldgs, None, None, None, None, B, h, w, Cout, h, w, c0 + c1, 1, 1, 1, 1, 0, 0, False, 1, 0, False)
        gxs = affine_act(gsm[:, :c0], _const_vec(float(k * k), c0, gy.device), None, None, 0)
        # gradient of the full-resolution segment: a Cout -> c1 1x1 convolution with W[:, c0:]^T
        w_f = weight.detach()[:, c0:].transpose(0, 1).contiguous()
        gxf, ldgf = alloc_nhwc(B, c1, H, W, gy.dtype, gy.device)
        _conv_launch(g, ldg, packed_weight(w_f, 'conv', 'fwd', gy.dtype), gxf, ldgf, None, None, None, None, B, H, W, Cout, H, W, c1,
                     1, 1, 1, 1, 0, 0, False, 1, 0, False)
        # weight gradient, segment by segment
        xsv, ldxs = aligned_view(xs)
        xfv, ldxf = aligned_view(xf)
        gw_s, _ = wgrad(xsv, ldxs, gpv, ldgp, weight.detach()[:, :c0], None, ConvSpec('conv', 1, 1, 1, 1, 0, 0, h, w))
        gw_f, _ = wgrad(xfv, ldxf, g, ldg, weight.detach()[:, c0:], None, ConvSpec('conv', 1, 1, 1, 1, 0, 0, H, W))
        gw = torch.cat([gw_s * float(k * k), gw_f], 1)
        return gxs, gxf, gw, None, None


def upcat_conv1x1(xs, xf, weight, act=0):
    """act(conv1x1(cat([nearest_upsample(xs to xf's size), xf], 1))); None when the fused kernel does not apply (f32 parity
    path, sizes that are not a power-of-two multiple): the caller then composes interpolate + concat + conv."""
    if xs.dtype != torch.bfloat16 or xf.dtype != torch.bfloat16 or weight.dim() != 4 or weight.shape[2:] != (1, 1):
        return None
    H, W, h, w = xf.shape[2], xf.shape[3], xs.shape[2], xs.shape[3]
    if h == 0 or H % h or W % w or H // h != W // w:
        return None
    k = H // h
    if k < 1 or k & (k - 1) or k > 64 or xs.shape[1] % 8:
        return None
    return _UpCatConv1x1Fn.apply(xs, xf, weight, k.bit_length() - 1, act)


def conv_same_geometry(H, W, k, stride, dil):
    """TF-'same' padding of conv2dSame (models/torch_model.py:276-281): output ceil(size/stride), extra pad bottom/right."""
    def one(size):
        out = -(-size // stride)
        total = max((out - 1) * stride - size + dil * (k - 1) + 1, 0)
        return out, total // 2
    Ho, pt = one(H)
    Wo, pl = one(W)
    return Ho, Wo, pt, pl


def deconv_same_geometry(H, W, k, dil):
    """Stride-1 ConvTranspose2dSame (models/torch_model.py:320-346): full transposed conv of size H+dil*(k-1),
    cropped from start = full//2 - H//2; as a correlation with flipped taps the top/left padding is dil*(k-1) - start."""
    D = dil * (k - 1)
    def one(size):
        start = (size + D) // 2 - size // 2
        return D - start
    return one(H), one(W)


def conv_spec(x, weight, kind='conv', stride=1, dilation=1, padding=0):
    """padding: int (symmetric, nn.Conv2d), 'same' (conv2dSame) or 'ctsame' (stride-1 ConvTranspose2dSame)."""
    B, C, H, W = x.shape
    kh, kw = weight.shape[2], weight.shape[3]
    if kind == 'deconv':
        if stride != 1 or padding != 'ctsame' or kh != kw:
            raise _lib.SdhipError("ConvTranspose2d is implemented for stride 1 with 'same' cropping only")
        pt, pl = deconv_same_geometry(H, W, kh, dilation)
        Ho, Wo = H, W
    elif padding == 'same':
        if kh != kw:
            raise _lib.SdhipError("'same' padding needs a square kernel")
        Ho, Wo, pt, pl = conv_same_geometry(H, W, kh, stride, dilation)
    else:
        pt, pl = (int(padding[0]), int(padding[1])) if isinstance(padding, (tuple, list)) else (int(padding), int(padding))
        Ho = (H + 2 * pt - dilation * (kh - 1) - 1) // stride + 1
        Wo = (W + 2 * pl - dilation * (kw - 1) - 1) // stride + 1
    return ConvSpec(kind, kh, kw, stride, dilation, pt, pl, Ho, Wo)


def conv2d(x, weight, bias=None, *, kind='conv', stride=1, dilation=1, padding=0, act=0):
    return _ConvFn.apply(x, weight, bias, conv_spec(x, weight, kind, stride, dilation, padding), act)


def conv_bn_act(x, weight, bn, *, kind='conv', stride=1, dilation=1, padding=0, act=0, residual=None, groups=1,
                in_slot=None, res_slot=None, in_bn=None, out_bn=None):
    """in_slot / res_slot (GradSlot, optional): the gradient of `residual` is parked in res_slot instead of being returned
    to autograd, and a gradient parked in in_slot is summed into this node's data gradient by its own launch.
    in_bn / out_bn (BNSlot, optional): x is the output of the convbn + ReLU node that owns in_bn / this node's output has
    exactly one consumer, which holds out_bn as its in_bn."""
    return _ConvBNActFn.apply(x, weight, bn.weight, bn.bias, residual, conv_spec(x, weight, kind, stride, dilation, padding),
                              bn, act, groups, in_slot, res_slot, in_bn, out_bn)


def bn_conv(x, stats, bn, weight, *, padding=0, groups=1):
    """conv(relu(bn(x))) with x's batch statistics `stats` ([groups][2][C] f64) supplied by the producer of x."""
    return _BNConvFn.apply(x, stats, weight, bn.weight, bn.bias, conv_spec(x, weight, 'conv', 1, 1, padding), bn, groups)


def bn_act(x, stats, bn, act=1, groups=1):
    return _BNActFn.apply(x, stats, bn.weight, bn.bias, bn, act, groups)


class _AffineActFn(torch.autograd.Function):
    """y = act(x*scale + shift) (+ residual) with CONSTANT scale/shift (no gradient to them)."""

    @staticmethod
    def forward(ctx, x, scale, shift, residual, act):
        _require_gpu(x)
        B, C, H, W = x.shape
        xv, ldx = nhwc_view(x)
        rv, ldr = nhwc_view(residual) if residual is not None else (None, 0)
        y = empty_nhwc(B, C, H, W, x.dtype, x.device)
        call("sdhip_affine_act", ptr(xv), ldx, ptr(y), C, ptr(rv), ldr, ptr(scale), ptr(shift), B * H * W, C, 1, act,
             dtype_code(x), stream_ptr())
        ctx.save_for_backward(xv, scale, shift)
        ctx.cfg = (ldx, act, residual is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        xv, scale, shift = ctx.saved_tensors
        ldx, act, has_res = ctx.cfg
        B, C, H, W = xv.shape
        g, ldg = nhwc_view(gy)
        gx = empty_nhwc(B, C, H, W, xv.dtype, xv.device)
        call("sdhip_affine_act_bwd", ptr(g), ldg, ptr(xv), ldx, ptr(gx), C, ptr(scale), ptr(shift), None, None, 1,
             B * H * W, C, 1, act, 0, 0, dtype_code(xv), stream_ptr())
        return gx, None, None, (gy if has_res else None), None


def affine_act(x, scale=None, shift=None, residual=None, act=0):
    return _AffineActFn.apply(x, scale, shift, residual, act)


# ============================================================================ pooling / resize / concat / broadcast product
class _MaxPool3s2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        B, C, H, W = x.shape
        xv, ldx = nhwc_view(x)
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = empty_nhwc(B, C, Ho, Wo, x.dtype, x.device)
        idx = torch.empty((B * Ho * Wo, C), dtype=torch.uint8, device=x.device)
        call("sdhip_maxpool3s2_fwd", ptr(xv), ldx, ptr(y), C, ptr(idx), B, H, W, C, dtype_code(x), stream_ptr())
        ctx.save_for_backward(idx)
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        (idx,) = ctx.saved_tensors
        B, C, H, W = ctx.shape
        g, ldg = nhwc_view(gy)
        gx = empty_nhwc(B, C, H, W, gy.dtype, gy.device)
        call("sdhip_maxpool3s2_bwd", ptr(g), ldg, ptr(idx), ptr(gx), C, B, H, W, C, dtype_code(gy), stream_ptr())
        return gx


def maxpool3s2(x):
    return _MaxPool3s2Fn.apply(x)


class _AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k):
        _require_gpu(x)
        B, C, H, W = x.shape
        xv, ldx = nhwc_view(x)
        y = empty_nhwc(B, C, H // k, W // k, x.dtype, x.device)
        call("sdhip_avgpool_fwd", ptr(xv), ldx, ptr(y), C, B, H, W, C, k, dtype_code(x), stream_ptr())
        ctx.cfg = (B, C, H, W, k)
        return y

    @staticmethod
    def backward(ctx, gy):
        B, C, H, W, k = ctx.cfg
        g, ldg = nhwc_view(gy)
        gx = empty_nhwc(B, C, H, W, gy.dtype, gy.device)
        call("sdhip_avgpool_bwd", ptr(g), ldg, ptr(gx), C, B, H, W, C, k, dtype_code(gy), stream_ptr())
        return gx, None


def avgpool(x, k):
    """k x k / stride k average pool.  Large windows are built as a chain of <= 8 x 8 pools (mean of equal-size
    means), which keeps every launch wide; nn.AvgPool2d(128) of a 128x256 map would otherwise be 2 serial threads."""
    while k > 8 and k % 2 == 0:
        step = 8 if k % 8 == 0 else 2
        x = _AvgPoolFn.apply(x, step)
        k //= step
    return _AvgPoolFn.apply(x, k) if k > 1 else x


_MODES = {'nearest': 0, 'bilinear': 1, 'bilinear_ac': 2}


class _ResizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo, mode, scale_h, scale_w, out):
        _require_gpu(x)
        B, C, H, W = x.shape
        xv, ldx = nhwc_view(x)
        if out is None:
            y, ldy = empty_nhwc(B, C, Ho, Wo, x.dtype, x.device), C
        else:
            y, ldy = out, out.stride(3)
        call("sdhip_resize_fwd", ptr(xv), ldx, ptr(y), ldy, B, H, W, C, Ho, Wo, mode, scale_h, scale_w, dtype_code(x), stream_ptr())
        ctx.cfg = (B, C, H, W, Ho, Wo, mode, scale_h, scale_w)
        return y

    @staticmethod
    def backward(ctx, gy):
        B, C, H, W, Ho, Wo, mode, scale_h, scale_w = ctx.cfg
        g, ldg = nhwc_view(gy)
        gx = empty_nhwc(B, C, H, W, gy.dtype, gy.device)
        tmp = torch.empty(B * Ho * W * C, dtype=torch.float32, device=gy.device)
        call("sdhip_resize_bwd", ptr(g), ldg, ptr(gx), C, ptr(tmp), B, H, W, C, Ho, Wo, mode, scale_h, scale_w,
             dtype_code(gy), stream_ptr())
        return gx, None, None, None, None, None, None


def interpolate(x, size=None, scale_factor=None, mode='nearest', align_corners=None):
    """F.interpolate for the modes on the hot path.  A same-size nearest / bilinear(align_corners=False) resize is the
    identity (source index == destination index, lambda == 0) and is skipped."""
    B, C, H, W = x.shape
    if size is not None:
        Ho, Wo = (size, size) if isinstance(size, int) else (int(size[0]), int(size[1]))
        sh = sw = 0.0
    else:
        Ho, Wo = int(H * scale_factor), int(W * scale_factor)
        sh = sw = 1.0 / float(scale_factor)   # ATen uses the user scale when recompute_scale_factor is unset
    if mode == 'bilinear' and align_corners:
        m = 2
    elif mode in ('nearest', 'bilinear'):
        m = _MODES[mode]
    else:
        raise _lib.SdhipError("interpolate mode %r is not on the hot path" % mode)
    if (Ho, Wo) == (H, W) and m != 2:
        return x
    return _ResizeFn.apply(x, Ho, Wo, m, sh, sw, None)


class _SplitBatchFn(torch.autograd.Function):
    """x -> (x[:B], x[B:]) for the two tower halves of a batched pass.  The forward is free (views); the backward writes the
    two incoming gradients into the halves of ONE NHWC buffer (autograd's own slice backward builds two zero-filled NCHW
    tensors, adds them and leaves the consumer a layout conversion: ~1 ms per step)."""

    @staticmethod
    def forward(ctx, x, B):
        ctx.set_materialize_grads(False)          # an unused half arrives as None (zeroed below), not as an NCHW zero tensor
        ctx.B = B
        ctx.meta = (tuple(x.shape), x.dtype, x.device)
        return x[:B], x[B:]

    @staticmethod
    def backward(ctx, ga, gb):
        (N, C, H, W), dtype, dev = ctx.meta
        B = ctx.B
        if ga is None and gb is None:
            return None, None
        g, ld = alloc_nhwc(N, C, H, W, dtype, dev)
        for part, lo, hi in ((ga, 0, B), (gb, B, N)):
            dst = g[lo:hi]
            if part is None:
                dst.zero_()
                continue
            pv, ldp = nhwc_view(part)
            call("sdhip_affine_act", ptr(pv), ldp, ptr(dst), ld, None, 0, None, None, (hi - lo) * H * W, C, 1, 0, dtype_code(pv),
                 stream_ptr())
        return g, None


def split_batch(x, B):
    return _SplitBatchFn.apply(x, B)


class _ConcatFn(torch.autograd.Function):
    """torch.cat(dim=1) as copies into channel slices of one NHWC slab; the backward is free (views)."""

    @staticmethod
    def forward(ctx, *xs):
        _require_gpu(*xs)
        B, _, H, W = xs[0].shape
        Ct = sum(t.shape[1] for t in xs)
        out, ldo = alloc_nhwc(B, Ct, H, W, xs[0].dtype, xs[0].device)
        off, offs = 0, []
        for t in xs:
            C = t.shape[1]
            tv, ld = nhwc_view(t)
            dst = out[:, off:off + C]
            call("sdhip_affine_act", ptr(tv), ld, ptr(dst), ldo, None, 0, None, None, B * H * W, C, 1, 0, dtype_code(t), stream_ptr())
            offs.append((off, C))
            off += C
        ctx.offs = offs
        return out

    @staticmethod
    def backward(ctx, g):
        return tuple(g[:, o:o + c] for o, c in ctx.offs)


def concat(xs):
    return _ConcatFn.apply(*xs)


class _MulBcastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, m):
        _require_gpu(a, m)
        B, C, H, W = a.shape
        av, lda = nhwc_view(a)
        mv, ldm = nhwc_view(m)
        y = empty_nhwc(B, C, H, W, a.dtype, a.device)
        call("sdhip_mul_bcast_fwd", ptr(av), lda, ptr(mv), ldm, ptr(y), C, B * H * W, C, dtype_code(a), stream_ptr())
        ctx.save_for_backward(av, mv)
        ctx.cfg = (lda, ldm)
        return y

    @staticmethod
    def backward(ctx, g):
        av, mv = ctx.saved_tensors
        lda, ldm = ctx.cfg
        B, C, H, W = av.shape
        gv, ldg = nhwc_view(g)
        ga = empty_nhwc(B, C, H, W, av.dtype, av.device)
        gm = empty_nhwc(B, 1, H, W, av.dtype, av.device)
        call("sdhip_mul_bcast_bwd", ptr(gv), ldg, ptr(av), lda, ptr(mv), ldm, ptr(ga), C, ptr(gm), 1, B * H * W, C,
             dtype_code(av), stream_ptr())
        return ga, gm


def mul_bcast(a, m):
    """a (B,C,H,W) * m (B,1,H,W)."""
    return _MulBcastFn.apply(a, m)


# ============================================================================ losses of the timed step
_lovasz_ws = {}


class _TrainLossFn(torch.autograd.Function):
    """total = CE(seg1) + CE(seg2) [+ Lovasz(seg2)] + L1(disp)  — the loss of the reference's training step
    (torch_implementation.py:279,293,304,325 with `-loss cross_entropy lovasz_loss`).  The gradients w.r.t. the
    three network outputs are produced in the same pass."""

    @staticmethod
    def forward(ctx, seg1, disp, seg2, seg_t, disp_t, use_lovasz, mask_invalid_disp=False, ignore_void=False):
        _require_gpu(seg1, disp, seg2, seg_t, disp_t)
        B, C, H, W = seg1.shape
        npix = B * H * W
        dt = dtype_code(seg1)
        loss = torch.zeros(1, dtype=torch.float64, device=seg1.device)
        tv, ldt = nhwc_view(seg_t)
        grads = []
        for s in (seg1, seg2):
            sv, ld = nhwc_view(s)
            g = empty_nhwc(B, C, H, W, s.dtype, s.device)
            call("sdhip_ce_loss", ptr(sv), ld, ptr(tv), ldt, ptr(g), C, ptr(loss), npix, C, 1.0, dt, stream_ptr())
            grads.append(g)
        dv, ldd = nhwc_view(disp)
        if ldd != 1:      # 1-channel map inside a padded pixel stride (direct conv output): densify
            dv = disp.contiguous()
        if not disp_t.is_contiguous():
            raise _lib.SdhipError("disparity target must be a dense (B,1,H,W) tensor")
        gd = torch.empty_like(dv)
        call("sdhip_l1_loss", ptr(dv), ptr(disp_t), ptr(gd), ptr(loss), npix, 1.0, int(mask_invalid_disp), dt, stream_ptr())
        if use_lovasz:   # added onto the CE gradient of seg2 in place
            s2, ld2 = nhwc_view(seg2)
            nbytes = _lib.lovasz_workspace_bytes(npix, C)
            key = (npix, C, str(seg1.device))
            ws = _lovasz_ws.get(key)
            if ws is None or ws.numel() < nbytes:
                ws = torch.empty(nbytes, dtype=torch.uint8, device=seg1.device)
                _lovasz_ws[key] = ws
            call("sdhip_lovasz_softmax", ptr(s2), ld2, ptr(tv), ldt, ptr(grads[1]), C, ptr(loss), npix, C, 1.0, ptr(ws),
                 ws.numel(), int(ignore_void), dt, stream_ptr())
        total = loss
        ctx.save_for_backward(grads[0], gd, grads[1])
        return total.float()

    @staticmethod
    def backward(ctx, g):
        g1, gd, g2 = ctx.saved_tensors
        return g1, gd, g2, None, None, None, None, None   # d(total)/d(total) is 1 in the training step


def train_loss(seg1, disp, seg2, seg_target, disp_target, use_lovasz=True, mask_invalid_disp=False, ignore_void=False):
    """seg_target: one-hot f32 (B,C,H,W); disp_target: f32 (B,1,H,W).  The two dataset rules of losses/multiLosses.py are
    opt-in, both off by default (roses / garden, `ignore=None`, :11-17):
      ignore_void       — cityscapes / kitti (:19-21, the 20th one-hot channel dropped, `ignore=19`): an all-zero target row
                          marks a void pixel, removed from the Lovasz term.  Off: its label is argmax = class 0 and it counts.
                          (Its cross-entropy term is zero under either rule: sum(-t * log_softmax) over a zero row.)
      mask_invalid_disp — disparities <= 0 are invalid (:134-141)."""
    return _TrainLossFn.apply(seg1, disp, seg2, seg_target, disp_target, use_lovasz, mask_invalid_disp, ignore_void)


# ============================================================================ dropout / global average pool
_rng = {"seed": None, "layers": 0}


_RNG_BASE, _RNG_RANK_STRIDE = 0x5DEECE66D, 0x9E3779B97F4A7C15 >> 1


def rng_seed_tensor(device):
    """Device-resident dropout seed.  train.TrainStep advances it once per step with a device-side add (captured into the
    step's hipGraph, so replays draw new masks); forward and backward of one step read the same value."""
    c = _ctx[0]
    if c is not None and c.seed is not None:                  # a TrainStep's own stream (steps must not reset each other)
        return c.seed
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:       # "cuda" and "cuda:0" must name the same tensor: a captured step
        device = torch.device("cuda", torch.cuda.current_device())   # keeps adding to the one it was recorded with
    if _rng["seed"] is None or _rng["seed"].device != device:
        _rng["seed"] = torch.full((1,), _RNG_BASE, dtype=torch.int64, device=device)
    return _rng["seed"]


def rng_seed_value(rank=0, base=_RNG_BASE):
    return (base + rank * _RNG_RANK_STRIDE) & 0x7FFFFFFFFFFFFFFF


def rng_reseed(device, rank=0, base=_RNG_BASE):
    """Restart the dropout stream (the module-level one, or the installed TrainStep's); ranks of a data-parallel job get
    disjoint streams (in place: a captured graph keeps reading the same tensor)."""
    rng_seed_tensor(device).fill_(rng_seed_value(rank, base))


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, layer_id):
        _require_gpu(x)
        B, C, H, W = x.shape
        xv, ld = nhwc_view(x)
        if ld != C:
            xv = xv.contiguous(memory_format=torch.channels_last)
        y = empty_nhwc(B, C, H, W, x.dtype, x.device)
        seed = rng_seed_tensor(x.device)
        call("sdhip_dropout", ptr(xv), ptr(y), ptr(seed), layer_id, x.numel(), p, dtype_code(x), stream_ptr())
        ctx.cfg = (p, layer_id)
        return y

    @staticmethod
    def backward(ctx, gy):
        p, layer_id = ctx.cfg
        B, C, H, W = gy.shape
        g, ld = nhwc_view(gy)
        if ld != C:
            g = g.contiguous(memory_format=torch.channels_last)
        gx = empty_nhwc(B, C, H, W, gy.dtype, gy.device)
        call("sdhip_dropout", ptr(g), ptr(gx), ptr(rng_seed_tensor(gy.device)), layer_id, gy.numel(), p, dtype_code(gy), stream_ptr())
        return gx, None, None


def dropout(x, p, training, layer_id):
    if not training or p == 0.0:
        return x
    return _DropoutFn.apply(x, float(p), int(layer_id))


# ============================================================================ HANet pieces (models_hanet/HANet.py:74-128)
class _RowPoolMaxFn(torch.autograd.Function):
    """nn.AdaptiveMaxPool2d((OH, 1)): (B,C,H,W) -> (B,C,OH,1)."""

    @staticmethod
    def forward(ctx, x, OH):
        _require_gpu(x)
        B, C, H, W = x.shape
        xv, ldx = nhwc_view(x)
        y, ldy = alloc_nhwc(B, C, OH, 1, x.dtype, x.device)
        idx = torch.empty((B, OH, C), dtype=torch.int32, device=x.device)
        call("sdhip_rowpool_max_fwd", ptr(xv), ldx, ptr(y), ldy, ptr(idx), B, H, W, C, OH, dtype_code(x), stream_ptr())
        ctx.save_for_backward(idx)
        ctx.cfg = (B, C, H, W, OH)
        return y

    @staticmethod
    def backward(ctx, gy):
        (idx,) = ctx.saved_tensors
        B, C, H, W, OH = ctx.cfg
        g, ldg = nhwc_view(gy)
        gx, ldgx = alloc_nhwc(B, C, H, W, gy.dtype, gy.device)
        call("sdhip_rowpool_max_bwd", ptr(g), ldg, ptr(idx), ptr(gx), ldgx, B, H, W, C, OH, dtype_code(gy), stream_ptr())
        return gx, None


def rowpool_max(x, OH):
    return _RowPoolMaxFn.apply(x, int(OH))


class _MulRowsFn(torch.autograd.Function):
    """a (B,C,H,W) * att (B,C,H,1): torch.mul(out, x1d.unsqueeze(3)) of models_hanet/HANet.py:112."""

    @staticmethod
    def forward(ctx, a, att):
        _require_gpu(a, att)
        B, C, H, W = a.shape
        if tuple(att.shape) != (B, C, H, 1):
            raise _lib.SdhipError("mul_rows: attention must be (B,C,H,1), got %s for %s" % (tuple(att.shape), tuple(a.shape)))
        av, lda = nhwc_view(a)
        tv, ldt = nhwc_view(att)
        y, ldy = alloc_nhwc(B, C, H, W, a.dtype, a.device)
        call("sdhip_mul_rows_fwd", ptr(av), lda, ptr(tv), ldt, ptr(y), ldy, B, H, W, C, dtype_code(a), stream_ptr())
        ctx.save_for_backward(av, tv)
        ctx.cfg = (lda, ldt)
        return y

    @staticmethod
    def backward(ctx, g):
        av, tv = ctx.saved_tensors
        lda, ldt = ctx.cfg
        B, C, H, W = av.shape
        gv, ldg = nhwc_view(g)
        ga, ldga = alloc_nhwc(B, C, H, W, av.dtype, av.device)
        gt, ldgt = alloc_nhwc(B, C, H, 1, av.dtype, av.device)
        call("sdhip_mul_rows_bwd", ptr(gv), ldg, ptr(av), lda, ptr(tv), ldt, ptr(ga), ldga, ptr(gt), ldgt, B, H, W, C,
             dtype_code(av), stream_ptr())
        return ga, gt


def mul_rows(a, att):
    return _MulRowsFn.apply(a, att)


class _DropoutChannelsFn(torch.autograd.Function):
    """nn.Dropout2d on a (B,C,L[,1]) row descriptor: whole (sample, channel) rows are dropped."""

    @staticmethod
    def forward(ctx, x, p, layer_id):
        _require_gpu(x)
        B, C, L, W = x.shape
        xv, ldx = nhwc_view(x)
        y, ldy = alloc_nhwc(B, C, L, W, x.dtype, x.device)
        call("sdhip_dropout_channels", ptr(xv), ldx, ptr(y), ldy, ptr(rng_seed_tensor(x.device)), layer_id, B, L * W, C, p,
             dtype_code(x), stream_ptr())
        ctx.cfg = (p, layer_id)
        return y

    @staticmethod
    def backward(ctx, gy):
        p, layer_id = ctx.cfg
        B, C, L, W = gy.shape
        g, ldg = nhwc_view(gy)
        gx, ldgx = alloc_nhwc(B, C, L, W, gy.dtype, gy.device)
        call("sdhip_dropout_channels", ptr(g), ldg, ptr(gx), ldgx, ptr(rng_seed_tensor(gy.device)), layer_id, B, L * W, C, p,
             dtype_code(gy), stream_ptr())
        return gx, None, None


def dropout_channels(x, p, training, layer_id):
    if not training or p == 0.0:
        return x
    return _DropoutChannelsFn.apply(x, float(p), int(layer_id))


def global_avg_pool(x):
    """nn.AdaptiveAvgPool2d((1,1)): gcd(H,W)-sized square average pools (equal windows, so the mean of the window means is
    the global mean), then the mean of the few remaining cells (a (B,C,h,w) tensor with h*w tiny)."""
    import math
    B, C, H, W = x.shape
    k = math.gcd(H, W)
    y = avgpool(x, k) if k > 1 else x
    return y.mean((2, 3), keepdim=True) if y.shape[2] * y.shape[3] > 1 else y


# ============================================================================ 3-D convolutions and PSMNet ops
def conv3d_spec(x, D, weight, stride=1, padding=1):
    """nn.Conv3d(k, stride, padding) over a volume held as (B*D, C, H, W) NHWC images."""
    Bimg, C, H, W = x.shape
    kd, kh, kw = weight.shape[2:]
    Do = (D + 2 * padding - kd) // stride + 1
    Ho = (H + 2 * padding - kh) // stride + 1
    Wo = (W + 2 * padding - kw) // stride + 1
    return ConvSpec('conv', kh, kw, stride, 1, padding, padding, Ho, Wo, D, Do, kd, stride, padding)


def conv3d_bn_act(x, D, weight, bn, stride=1, padding=1, act=0, residual=None, groups=1, in_slot=None, res_slot=None):
    """in_slot / res_slot: see conv_bn_act (GradSlot)."""
    spec = conv3d_spec(x, D, weight, stride, padding)
    return _ConvBNActFn.apply(x, weight, bn.weight, bn.bias, residual, spec, bn, act, groups, in_slot, res_slot), spec.Do


def conv3d(x, D, weight, stride=1, padding=1):
    spec = conv3d_spec(x, D, weight, stride, padding)
    return _ConvFn.apply(x, weight, None, spec, 0), spec.Do


class _StuffFn(torch.autograd.Function):
    """Zero insertion: y[n, d*sd, h*s, w*s] = x[n,d,h,w] (extent (D-1)*sd+1 ...); backward is the strided gather."""

    @staticmethod
    def forward(ctx, x, D, sd, s):
        _require_gpu(x)
        Bimg, C, H, W = x.shape
        B = Bimg // D
        xv, ld = nhwc_view(x)
        Ds, Hs, Ws = (D - 1) * sd + 1, (H - 1) * s + 1, (W - 1) * s + 1
        y = empty_nhwc(B * Ds, C, Hs, Ws, x.dtype, x.device)
        call("sdhip_stuff", ptr(xv), ld, ptr(y), C, B, D, H, W, C, sd, s, 1, dtype_code(x), stream_ptr())
        ctx.cfg = (B, D, H, W, C, sd, s)
        return y

    @staticmethod
    def backward(ctx, gy):
        B, D, H, W, C, sd, s = ctx.cfg
        g, ld = nhwc_view(gy)
        gx = empty_nhwc(B * D, C, H, W, gy.dtype, gy.device)
        call("sdhip_stuff", ptr(g), ld, ptr(gx), C, B, D, H, W, C, sd, s, 0, dtype_code(gy), stream_ptr())
        return gx, None, None, None


# Sub-pixel decomposition of ConvTranspose3d(k=3, s=2, p=1, output_padding=1): per axis, output 2m takes input m with tap 1;
# output 2m+1 takes inputs m, m+1 with taps 2, 0.  Phase (pd, ph, pw) is a stride-1 correlation with a (1+pd)x(1+ph)x(1+pw)
# sub-kernel; the 27 taps, ordered phase by phase (input offset major), index the flat 3x3x3 kernel as follows.
_PHASE_TAPS = ([1], [2, 0])
_PHASES = [(pd, ph, pw) for pd in (0, 1) for ph in (0, 1) for pw in (0, 1)]
_PHASE_INDEX = [a * 9 + b * 3 + c for (pd, ph, pw) in _PHASES for a in _PHASE_TAPS[pd] for b in _PHASE_TAPS[ph] for c in _PHASE_TAPS[pw]]
_phase_index_dev = {}


def _phase_rows(weight, dt):
    """Single-tap pack rows (src element offset, dst element offset, M, K, 1, stride_m, stride_k, 0) that build the eight packed
    sub-kernels of a (Cin, Cout, 3, 3, 3) transposed-convolution weight straight from the parameter, one buffer behind the
    other, and [(element offset, elements)] of the eight sub-kernels.  Needs one channel chunk (Cin <= 64 bf16 / 32 f32): a
    tap's [Mpad][CK] block is then contiguous in the packed [kd][q][t][m][c] layout."""
    Cin, Cout = weight.shape[0], weight.shape[1]
    rows, spans, base = [], [], 0
    for (pd, ph, pw) in _PHASES:
        nd, T = 1 + pd, (1 + ph) * (1 + pw)
        per = _lib.packed_elems(Cout, Cin, T, dt)
        blk = per // T                                   # one tap's [Mpad][CK] block (single channel chunk)
        for kdi, a in enumerate(_PHASE_TAPS[pd]):
            t = 0
            for b_ in _PHASE_TAPS[ph]:
                for c_ in _PHASE_TAPS[pw]:
                    rows.append((a * 9 + b_ * 3 + c_, base + kdi * per + t * blk, Cout, Cin, 1, 27, Cout * 27, 0))
                    t += 1
        spans.append((base, per * nd))
        base += per * nd
    return rows, spans


def _phase_packs(weight, dtype):
    """The eight packed sub-kernels of a (Cin, Cout, 3, 3, 3) ConvTranspose3d weight (equally: of a stride-2 Conv3d weight
    (Cout', Cin', 3, 3, 3) read as its adjoint), taps ordered phase by phase, input offset major.  One channel chunk: cached
    per parameter like ops.packed_weight and, inside a training step, packed by the step's ONE batched launch (27 single-tap
    rows per weight in the descriptor table).  Wider weights: a gather reorders the taps into a (Cout, Cin, 27) buffer, from
    which every (phase, depth tap) is packed with its own strides."""
    Cin, Cout = weight.shape[0], weight.shape[1]
    dev = weight.device
    dt = _lib.BF16 if dtype == torch.bfloat16 else _lib.F32
    if Cin <= (64 if dtype == torch.bfloat16 else 32) and weight.is_contiguous():
        ent = _cache_entry(weight)
        key = ('phase', 'fwd', dtype)
        hit = ent.get(key)
        c = _ctx[0]
        rows, spans = _phase_rows(weight, dt)
        ver = (weight._version, _pack_generation[0])
        fresh = hit is not None and ((c is not None and c.frozen_pack) or (hit[0] == ver and not torch.cuda.is_current_stream_capturing()))
        if fresh:
            buf = hit[1]
        else:
            buf = torch.empty(spans[-1][0] + spans[-1][1], dtype=dtype, device=dev)
            es = buf.element_size()
            w = weight.detach()
            for src_off, dst_off, M, K, T, sm, sk, flip in rows:
                call("sdhip_conv_pack_weights", ctypes_ptr(w.data_ptr() + 4 * src_off), ctypes_ptr(buf.data_ptr() + es * dst_off),
                     M, K, T, sm, sk, flip, dt, stream_ptr())
            ent[key] = (ver, buf)
        return [buf[o:o + n] for o, n in spans], buf
    idx = _phase_index_dev.get(str(dev))
    if idx is None:
        idx = _phase_index_dev[str(dev)] = torch.tensor(_PHASE_INDEX, dtype=torch.int64, device=dev)
    wre = weight.detach().reshape(Cin, Cout, 27).index_select(2, idx).transpose(0, 1).contiguous()      # (Cout, Cin, 27)
    packs, off = [], 0
    for (pd, ph, pw) in _PHASES:
        nd, T = 1 + pd, (1 + ph) * (1 + pw)
        per = _lib.packed_elems(Cout, Cin, T, dt)
        buf = torch.empty(per * nd, dtype=dtype, device=dev)
        es = buf.element_size()
        for kdi in range(nd):
            call("sdhip_conv_pack_weights", ctypes_ptr(wre.data_ptr() + 4 * (off + kdi * T)), ctypes_ptr(buf.data_ptr() + es * per * kdi),
                 Cout, Cin, T, Cin * 27, 27, 0, dt, stream_ptr())
        packs.append(buf)
        off += nd * T
    return packs, wre


class _Deconv3dS2BNActFn(torch.autograd.Function):
    """y = act(BatchNorm3d(ConvTranspose3d(k=3, s=2, p=1, op=1)(x))) (+ residual), models_psmnet/stackhourglass.py:25-29,42-48.
    Forward: eight sub-pixel phases over the un-stuffed volume (27 taps per input voxel, not 8 x 27 over a zero-stuffed
    one), their statistics summed by the conv epilogues.  Backward: the adjoint of a transposed convolution is the
    stride-2 convolution with the same weight, so the data gradient is a strided forward convolution of dy and the weight
    gradient the strided weight-gradient kernel with the roles of x and dy exchanged — no stuffing either."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, D, bn, act, groups):
        _require_gpu(x, weight)
        Bimg, Cin, H, W = x.shape
        Btrue = Bimg // D
        Cout = weight.shape[1]
        xv, ldx = aligned_view(x)
        Bo = Btrue * 2 * D
        yraw, ldr_ = alloc_nhwc(Bo, Cout, 2 * H, 2 * W, x.dtype, x.device)
        train = bn.training
        ws = _zeros((NREP, groups, 2, Cout), torch.float64, x.device)[0] if train else None
        packs, keep = _phase_packs(weight, x.dtype)
        for (pd, ph, pw), wp in zip(_PHASES, packs):
            call("sdhip_conv2d_fwd_phase", ptr(xv), ptr(wp), ptr(yraw), ptr(ws), ws.stride(-2) if ws is not None else 0, NREP if ws is not None else 1,
                 Btrue, H, W, Cin, ldx, Cout, ldr_, 1 + ph, 1 + pw, D, 1 + pd, groups, pd, ph, pw, dtype_code(x), stream_ptr())
        count = (Bo // groups) * 4 * H * W
        rv, ldr = nhwc_view(residual) if residual is not None else (None, 0)
        y, ldy = alloc_nhwc(Bo, Cout, 2 * H, 2 * W, x.dtype, x.device)
        if train and _fused_bn():
            scale, shift, mean, invstd = [torch.empty((groups, Cout), dtype=torch.float32, device=x.device) for _ in range(4)]
            _bn_track(bn, groups)
            parallel.all_reduce_sum_(ws)
            call("sdhip_affine_act_bn", ptr(yraw), ldr_, ptr(y), ldy, ptr(rv), ldr, ptr(ws), ws.stride(-2), NREP, ptr(bn.weight),
                 ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var), ptr(scale), ptr(shift), ptr(mean), ptr(invstd),
                 Bo * 4 * H * W, Cout, groups, float(parallel.global_count(count)), float(bn.eps),
                 float(0.1 if bn.momentum is None else bn.momentum), act, dtype_code(x), stream_ptr())
        else:
            scale, shift, mean, invstd = _bn_finalize(ws, NREP, bn, count, groups)
            call("sdhip_affine_act", ptr(yraw), ldr_, ptr(y), ldy, ptr(rv), ldr, ptr(scale), ptr(shift), Bo * 4 * H * W,
                 Cout, groups, act, dtype_code(x), stream_ptr())
        ctx.cfg = (D, act, groups, ldx, ldr_, count, train, residual is not None)
        ctx.save_for_backward(xv, weight, gamma, beta, yraw, scale, shift, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        xv, weight, gamma, beta, yraw, scale, shift, mean, invstd = ctx.saved_tensors
        D, act, groups, ldx, ldraw, count, train, has_res = ctx.cfg
        Bimg, Cin, H, W = xv.shape
        Btrue = Bimg // D
        Bo, Cout = yraw.shape[0], yraw.shape[1]
        Ho, Wo = 2 * H, 2 * W
        npix = Bo * Ho * Wo
        dt = dtype_code(xv)
        g, ldg = nhwc_view(gy)
        graw, ldgr = alloc_nhwc(Bo, Cout, Ho, Wo, xv.dtype, xv.device)
        if train and act in (0, 1, 2) and _fused_bn():
            dgamma, dbeta = bn_backward_two_phase(g, ldg, yraw, ldraw, graw, ldgr, scale, shift, mean, invstd, gamma, beta, npix,
                                                  Cout, groups, act, parallel.global_count(count), dt)
        elif train and act in (0, 1, 2):
            dgamma, dbeta, dS = _bn_backward(g, ldg, yraw, ldraw, None, 0, scale, shift, mean, invstd, gamma, npix, Cout,
                                             groups, act, count, True, dt, beta=beta)
            call("sdhip_bn_bwd_apply", ptr(g), ldg, ptr(yraw), ldraw, ptr(graw), ldgr, ptr(scale), ptr(shift), ptr(dS), Cout,
                 npix, Cout, groups, act, dt, stream_ptr())
        else:
            dgamma, dbeta, dS = _bn_backward(g, ldg, yraw, ldraw, graw, ldgr, scale, shift, mean, invstd, gamma, npix, Cout,
                                             groups, act, count, train, dt, beta=beta)
        # adjoint of the transposed convolution: Conv3d(weight viewed as (out = Cin, in = Cout), stride 2, padding 1) of graw
        spec = ConvSpec('conv', 3, 3, 2, 1, 1, 1, H, W, 2 * D, D, 3, 2, 1)
        gx = None
        if ctx.needs_input_grad[0]:
            gx, ldgx = alloc_nhwc(Bimg, Cin, H, W, xv.dtype, xv.device)
            _conv_launch(graw, ldgr, packed_weight(weight, 'conv', 'fwd', xv.dtype), gx, ldgx, None, None, None, None, Btrue, Ho, Wo, Cout,
                         H, W, Cin, 3, 3, 2, 1, 1, 1, False, 1, 0, False, 1, spec.depth())
        gw = None
        if ctx.needs_input_grad[1]:
            gw, _ = wgrad(graw, ldgr, xv, ldx, weight, None, spec)       # "input" = dy, "output gradient" = x: same (Cin, Cout, 3,3,3) layout
        return gx, gw, dgamma, dbeta, (gy if has_res else None), None, None, None, None


def deconv3d_s2_bn_act(x, D, weight, bn, act=0, residual=None, groups=1):
    """nn.ConvTranspose3d(k=3, stride=2, padding=1, output_padding=1) + BatchNorm3d (models_psmnet/stackhourglass.py:25-29):
    output extent exactly 2x; sub-pixel phases forward, strided convolution kernels backward (_Deconv3dS2BNActFn)."""
    return _Deconv3dS2BNActFn.apply(x, weight, bn.weight, bn.bias, residual, D, bn, act, groups), 2 * D


class _CostVolumeFn(torch.autograd.Function):
    """Concatenation cost volume of PSMNet (models_psmnet/stackhourglass.py:110-119) as (B*D, 2C, H, W) images."""

    @staticmethod
    def forward(ctx, left, right, D):
        _require_gpu(left, right)
        B, C, H, W = left.shape
        lv, ldl = nhwc_view(left)
        rv, ldr = nhwc_view(right)
        if ldl != ldr:
            lv = lv.contiguous(memory_format=torch.channels_last); rv = rv.contiguous(memory_format=torch.channels_last)
            ldl = C
        vol = empty_nhwc(B * D, 2 * C, H, W, left.dtype, left.device)
        call("sdhip_cost_volume_fwd", ptr(lv), ptr(rv), ldl, ptr(vol), B, D, H, W, C, dtype_code(left), stream_ptr())
        ctx.cfg = (B, C, H, W, D)
        return vol

    @staticmethod
    def backward(ctx, g):
        B, C, H, W, D = ctx.cfg
        gv, ld = nhwc_view(g)
        if ld != 2 * C:
            gv = gv.contiguous(memory_format=torch.channels_last)
        gl = empty_nhwc(B, C, H, W, g.dtype, g.device)
        gr = empty_nhwc(B, C, H, W, g.dtype, g.device)
        call("sdhip_cost_volume_bwd", ptr(gv), ptr(gl), ptr(gr), C, B, D, H, W, C, dtype_code(g), stream_ptr())
        return gl, gr, None


def cost_volume(left, right, D):
    return _CostVolumeFn.apply(left, right, D)


class _SoftArgminFn(torch.autograd.Function):
    """trilinear upsample -> softmax over disparity -> expectation (stackhourglass.py:138-155, submodule.py:56-64)."""

    @staticmethod
    def forward(ctx, cost, D4, maxdisp, H, W):
        _require_gpu(cost)
        Bimg, C, H4, W4 = cost.shape
        if C != 1:
            raise _lib.SdhipError("soft-argmin expects a 1-channel cost volume")
        B = Bimg // D4
        c = cost.contiguous()
        pred = torch.empty((B, H, W), dtype=cost.dtype, device=cost.device)
        # (log-sum-exp, pred) per pixel: the backward pass starts from them instead of repeating the two softmax passes
        stats = torch.empty((B * H * W, 2), dtype=torch.float32, device=cost.device) if (maxdisp == 4 * D4 and ctx.needs_input_grad[0]) else None
        call("sdhip_softargmin_fwd", ptr(c), ptr(pred), ptr(stats), B, D4, H4, W4, maxdisp, H, W, dtype_code(cost), stream_ptr())
        ctx.save_for_backward(c, stats)
        ctx.cfg = (B, D4, H4, W4, maxdisp, H, W)
        return pred

    @staticmethod
    def backward(ctx, g):
        c, stats = ctx.saved_tensors
        B, D4, H4, W4, maxdisp, H, W = ctx.cfg
        gc = torch.empty_like(c)
        nws = _lib._lib.sdhip_softargmin_bwd_workspace_floats(B, D4, H4, W4, maxdisp, H, W)
        tmp = torch.empty(nws, dtype=torch.float32, device=c.device)
        call("sdhip_softargmin_bwd", ptr(c), ptr(g.contiguous()), ptr(stats), ptr(gc), ptr(tmp), nws, B, D4, H4, W4, maxdisp, H, W,
             dtype_code(c), stream_ptr())
        return gc, None, None, None, None


def soft_argmin(cost, D4, maxdisp, H, W):
    return _SoftArgminFn.apply(cost, D4, maxdisp, H, W)


def relu(x):
    return affine_act(x, None, None, None, 1)


def add(x, y):
    return affine_act(x, None, None, y, 0)


class _MeanL1Fn(torch.autograd.Function):
    """mean over the predictions of mean |pred - target| — the (build-defined) PSMNet training loss: the reference has no
    PSMNet training harness; value and gradients in one pass per prediction."""

    @staticmethod
    def forward(ctx, target, *preds):
        _require_gpu(target, *preds)
        loss = torch.zeros(1, dtype=torch.float64, device=target.device)
        t = target.contiguous().float()
        grads = []
        for p in preds:
            pv = p.contiguous()
            g = torch.empty_like(pv)
            call("sdhip_l1_loss", ptr(pv), ptr(t), ptr(g), ptr(loss), pv.numel(), 1.0 / len(preds), 0, dtype_code(pv), stream_ptr())
            grads.append(g)
        ctx.save_for_backward(*grads)
        return loss.float()

    @staticmethod
    def backward(ctx, g):
        return (None,) + tuple(ctx.saved_tensors)


def mean_l1_loss(preds, target):
    return _MeanL1Fn.apply(target, *preds)


# ============================================================================ dsnet extras
class _LogSoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        B, C, H, W = x.shape
        xv, ld = nhwc_view(x)
        y = empty_nhwc(B, C, H, W, x.dtype, x.device)
        call("sdhip_log_softmax_fwd", ptr(xv), ld, ptr(y), C, B * H * W, C, dtype_code(x), stream_ptr())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        B, C, H, W = y.shape
        g, ld = nhwc_view(gy)
        gx = empty_nhwc(B, C, H, W, y.dtype, y.device)
        call("sdhip_log_softmax_bwd", ptr(g), ld, ptr(y), C, ptr(gx), C, B * H * W, C, dtype_code(y), stream_ptr())
        return gx


def log_softmax(x):
    return _LogSoftmaxFn.apply(x)


_consts = {}


def _const_vec(v, n, device):
    key = (float(v), n, str(device))
    if key not in _consts:
        _consts[key] = torch.full((1, n), float(v), dtype=torch.float32, device=device)
    return _consts[key]


def axpby(a, x, b, y):
    """a*x + b*y (the 0.9/0.1 and 0.8/0.2 head blends of models/dsnet_t2.py:272,308)."""
    C = x.shape[1]
    t = affine_act(y, _const_vec(b, C, x.device), None, None, 0)
    return affine_act(x, _const_vec(a, C, x.device), None, t, 0)


def deconv_same_strided_spec(x, weight, stride):
    """ConvTranspose2dSame with stride > 1 (models/torch_model.py:320-346): full transposed conv of extent (H-1)*s+k,
    cropped from start = full//2 - (s*H)//2 to s*H.  Run as a stride-1 correlation with flipped taps over the
    zero-stuffed input: top/left padding (k-1) - start."""
    B, C, H, W = x.shape
    k = weight.shape[2]

    def one(size):
        full, target = (size - 1) * stride + k, size * stride
        h, oh = full // 2, target // 2
        start = h - (oh if h - oh >= 0 else h)
        return (k - 1) - start, target
    pt, Ho = one(H)
    pl, Wo = one(W)
    return ConvSpec('deconv', k, k, 1, 1, pt, pl, Ho, Wo)


def deconv2d_strided(x, weight, bias, stride, bn=None, act=0, residual=None, groups=1):
    xs = _StuffFn.apply(x, 1, 1, stride)
    spec = deconv_same_strided_spec(x, weight, stride)
    if bn is None:
        return _ConvFn.apply(xs, weight, bias, spec, act)
    return _ConvBNActFn.apply(xs, weight, bn.weight, bn.bias, residual, spec, bn, act, groups)
