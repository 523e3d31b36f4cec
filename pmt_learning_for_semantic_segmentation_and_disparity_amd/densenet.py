"""DenseNet-121 feature tower of the reference (models/densenet.py), MI355X-native.

Same module tree / state_dict keys as the reference (`conv0`, `features.norm0`,
`denseblock.{0..6}.denselayerN.{norm1,conv1,norm2,conv2}`, `norm5`, `classifier`),
but a dense block runs as ONE autograd node over a pre-allocated NHWC channel
slab:

  * no torch.cat: every layer's 3x3 conv writes its 32 new channels straight into
    its channel slice of the slab (the reference re-concatenates all previous
    features for every layer, models/densenet.py:41-45,110-116);
  * no stand-alone BatchNorm pass: norm1/norm2 + ReLU are the fused input
    prologue of the 1x1 / 3x3 conv kernels, and the batch statistics they need
    are summed in the epilogue of the conv that produced the channels.  All
    layers that read a slab channel share its statistics (they only differ in
    gamma/beta), so each channel's sum / sum-of-squares is computed once;
  * left and right image run through the tower in one launch sequence as two
    statistics groups, which reproduces the reference's two separate calls
    (separate batch statistics, running stats updated left then right).
"""
from collections import OrderedDict

import os

import torch
import torch.nn as nn

from . import ops
from . import _lib as _lib_mod
from ._lib import call, dtype_code, ptr, stream_ptr


NREP = ops.NREP


def _finalize(stats, nrep, bn, count, groups, training, synced=False):
    return ops._bn_finalize(stats if training else None, nrep, bn, count, groups, synced=synced)


def _fold(ws, S_slice, groups, C, Ct):
    """Fold conv-epilogue replicas (summed over ranks when data parallel) into a channel slice of the slab statistics."""
    ws, nrep = ops._sync_stats(ws, ws.shape[0])
    call("sdhip_stats_replica_sum", ptr(ws), ptr(S_slice), nrep, groups, C, C, Ct, stream_ptr())


def _fold_finalize(ws, c_new0, Cn, S, Ct, bn, C, count, groups, synced=False):
    """sdhip_bn_fold_finalize: (scale, shift, mean, invstd) of `bn` over the first C slab channels, folding the replicas `ws`
    of the Cn newest channels (at c_new0) into the slab statistics S on the way (single GPU: no exchange in between)."""
    dev = bn.weight.device
    out = [torch.empty((groups, C), dtype=torch.float32, device=dev) for _ in range(4)]
    ops._bn_track(bn, groups)
    mom = 0.1 if bn.momentum is None else bn.momentum
    if not synced:
        ops.parallel.all_reduce_sum_(ws)          # sync-BN: the newest channels' replica sums become global in place
    call("sdhip_bn_fold_finalize", ptr(ws), ws.shape[0], ws.stride(-2), c_new0, Cn, ptr(S), Ct, ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean),
         ptr(bn.running_var), ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]), C, groups, float(ops.parallel.global_count(count)),
         float(bn.eps), float(mom), stream_ptr())
    return out


def _wgrad(x, ldx, dy, lddy, weight, B, H, W, Cin, Cout, k, pad, in_scale, in_shift, groups, dt):
    """dW of a stride-1 nn.Conv2d (weight (Cout,Cin,k,k)) whose input is relu(prologue(x)); x is a (B,Cin,H,W) view."""
    spec = ops.ConvSpec('conv', k, k, 1, 1, pad, pad, H, W)
    gw, _ = ops.wgrad(x[:, :Cin], ldx, dy, lddy, weight, None, spec, in_scale, in_shift, True, groups)
    return gw


# maps up to this many pixels: norm1's backward rides in the 1x1 data gradient (blocks 2-4 at 256x512; measured 16384 / 32768 /
# 131072 -> 25.36 / 25.29 / 25.30 ms: on the 64x128 maps of block 1 the streaming GEMM + separate pass is as fast)
FUSE1_MAX_PIX = _lib_mod.TUNE_FUSE1_MAX_PIX


PRO_MAX_PIX = _lib_mod.TUNE_PRO_MAX_PIX    # maps up to this many pixels: norm2 is finalized inside conv2's launch


class _DenseBlockFn(torch.autograd.Function):
    """(slab, slab statistics) = dense_block(x0); see the module docstring."""

    @staticmethod
    def forward(ctx, x0, block, groups, *params):
        layers = list(block.values())
        L = len(layers)
        B, C0, H, W = x0.shape
        growth = layers[0].conv2.out_channels
        mid = layers[0].conv1.out_channels
        Ct = C0 + L * growth
        dev, dtype = x0.device, x0.dtype
        dt = dtype_code(x0)
        training = block.training
        npix = B * H * W
        count = (B // groups) * H * W
        slab = ops.empty_nhwc(B, Ct, H, W, dtype, dev)
        xv, ldx = ops.nhwc_view(x0)
        call("sdhip_affine_act", ptr(xv), ldx, ptr(slab), Ct, None, 0, None, None, npix, C0, 1, 0, dt, stream_ptr())
        S = torch.zeros((groups, 2, Ct), dtype=torch.float64, device=dev)
        if training:   # statistics of the incoming features (the producer was a pool, not a conv epilogue)
            ws0 = torch.empty((NREP, groups, 2, C0), dtype=torch.float64, device=dev)
            call("sdhip_channel_stats", ptr(slab), Ct, ptr(ws0), C0, NREP, npix, C0, groups, 1, dt, stream_ptr())
            _fold(ws0, S[:, :, :C0], groups, C0, Ct)
        saved = []
        pending = None      # (replicas of the previous layer's new channels, their channel offset): folded by the next finalize
        fuse = training        # (also under data parallelism: the exchanges are in-place all-reduces between producer and consumer)
        gcount = ops.parallel.global_count(count)
        small = fuse and npix <= PRO_MAX_PIX and not _lib_mod.DIAG_NO_BNPRO   # finalize kernels folded into the convolutions
        nrep3 = 2 if small else NREP                      # replicas of a layer's output statistics (read back by <= 4 at a time)
        for li, layer in enumerate(layers):
            Cin = C0 + li * growth
            w1 = ops.packed_weight(layer.conv1.weight, 'conv', 'fwd', dtype)
            y1 = ops.empty_nhwc(B, mid, H, W, dtype, dev)
            # norm2's finalize inside conv2's launch (sdhip_conv2d_fwd_bnpro): one dependent node less per layer on the maps
            # where the tower is a latency chain; its table is built from <= 4 statistics replicas, so conv1 spreads its
            # epilogue atomics over fewer of them there (few workgroups: little contention)
            pro2 = small and mid <= 1024
            nrep2 = 2 if pro2 else NREP
            S2 = ops._zeros((nrep2, groups, 2, mid), torch.float64, dev)[0] if training else None
            done1 = False
            if small and Cin <= 1024:
                # ... and norm1's finalize (with the fold of the previous layer's statistics into the slab's) inside conv1's
                bn1 = layer.norm1
                sc1, sh1, mu1, iv1 = [torch.empty((groups, Cin), dtype=torch.float32, device=dev) for _ in range(4)]
                pw, pc0 = (pending[:2] if pending is not None else (None, 0))
                if pw is not None and not pending[2]:
                    ops.parallel.all_reduce_sum_(pw)     # the previous layer's output statistics (folded into the slab's by this launch)
                    pending = (pw, pc0, True)
                rc = _lib_mod._lib.sdhip_conv2d_fwd_bnpro(
                    ptr(slab), ptr(w1), ptr(y1), ptr(S2), S2.stride(-2), nrep2, ptr(S), Ct, 1,
                    ptr(pw), pw.stride(-2) if pw is not None else 0, pw.shape[0] if pw is not None else 0, pc0, growth if pw is not None else 0,
                    ptr(bn1.weight), ptr(bn1.bias), ptr(bn1.running_mean), ptr(bn1.running_var), ptr(sc1), ptr(sh1), ptr(mu1), ptr(iv1),
                    float(gcount), float(bn1.eps), float(0.1 if bn1.momentum is None else bn1.momentum),
                    B, H, W, Cin, Ct, H, W, mid, mid, 1, 1, 0, 0, groups, dt, stream_ptr())
                if rc == 0:
                    ops._bn_track(bn1, groups)
                    done1, pending = True, None
                elif rc != _lib_mod.ERR_UNSUPPORTED:
                    raise _lib_mod.SdhipError("sdhip_conv2d_fwd_bnpro failed (%d): %s" % (rc, _lib_mod._lib.sdhip_last_error().decode()))
            if not done1:
                if pending is not None:
                    # ONE launch: fold the previous layer's statistics into the slab AND finalize this layer's norm1
                    sc1, sh1, mu1, iv1 = _fold_finalize(pending[0], pending[1], growth, S, Ct, layer.norm1, Cin, count, groups, pending[2])
                    pending = None
                else:
                    sc1, sh1, mu1, iv1 = _finalize(S[:, :, :Cin], 1, layer.norm1, count, groups, training, synced=True)
                ops._conv_launch(slab, Ct, w1, y1, mid, None, sc1, sh1, S2, B, H, W, Cin, H, W, mid, 1, 1, 1, 1, 0, 0,
                                 True, groups, 0, False, nrep2)
            w2 = ops.packed_weight(layer.conv2.weight, 'conv', 'fwd', dtype)
            S3 = ops._zeros((nrep3, groups, 2, growth), torch.float64, dev)[0] if training else None
            done2 = s2_synced = False
            if pro2:
                bn2 = layer.norm2
                sc2, sh2, mu2, iv2 = [torch.empty((groups, mid), dtype=torch.float32, device=dev) for _ in range(4)]
                ops.parallel.all_reduce_sum_(S2)         # sync-BN: conv1's epilogue sums, in place, before conv2 finalizes them
                s2_synced = True
                rc = _lib_mod._lib.sdhip_conv2d_fwd_bnpro(
                    ptr(y1), ptr(w2), ptr(slab[:, Cin:Cin + growth]), ptr(S3), S3.stride(-2), nrep3, ptr(S2), S2.stride(-2), nrep2,
                    None, 0, 0, 0, 0, ptr(bn2.weight), ptr(bn2.bias), ptr(bn2.running_mean), ptr(bn2.running_var), ptr(sc2), ptr(sh2), ptr(mu2), ptr(iv2),
                    float(gcount), float(bn2.eps), float(0.1 if bn2.momentum is None else bn2.momentum),
                    B, H, W, mid, mid, H, W, growth, Ct, 3, 3, 1, 1, groups, dt, stream_ptr())
                if rc == 0:
                    ops._bn_track(bn2, groups)
                    done2 = True
                elif rc != _lib_mod.ERR_UNSUPPORTED:
                    raise _lib_mod.SdhipError("sdhip_conv2d_fwd_bnpro failed (%d): %s" % (rc, _lib_mod._lib.sdhip_last_error().decode()))
            if not done2:
                sc2, sh2, mu2, iv2 = _finalize(S2, nrep2, layer.norm2, count, groups, training, synced=s2_synced)
                ops._conv_launch(y1, mid, w2, slab[:, Cin:Cin + growth], Ct, None, sc2, sh2, S3, B, H, W, mid, H, W, growth,
                                 3, 3, 1, 1, 1, 1, True, groups, 0, False, nrep3)
            if training:   # fold the replicas into this layer's slice of the slab statistics
                if fuse and li + 1 < L:
                    pending = (S3, Cin, False)   # ... together with the next layer's norm1 finalize (False: not yet summed over ranks)
                else:
                    _fold(S3, S[:, :, Cin:Cin + growth], groups, growth, Ct)
            saved.append((y1, sc1, sh1, mu1, iv1, sc2, sh2, mu2, iv2))
        ctx.block, ctx.groups, ctx.saved, ctx.slab = block, groups, saved, slab
        ctx.geom = (B, C0, H, W, growth, mid, Ct, training, count)
        return slab, S

    @staticmethod
    def backward(ctx, g_slab_in, gS_in):
        ops.overlap_point()      # the queued weight gradients may run beside this block's chain of small kernels
        layers = list(ctx.block.values())
        groups, slab = ctx.groups, ctx.slab
        B, C0, H, W, growth, mid, Ct, training, count = ctx.geom
        dev, dtype = slab.device, slab.dtype
        dt = dtype_code(slab)
        npix = B * H * W
        st = stream_ptr()
        # accumulators: gradient w.r.t. the slab and w.r.t. its statistics
        g_slab = ops.empty_nhwc(B, Ct, H, W, dtype, dev)
        gv, ldg = ops.nhwc_view(g_slab_in)
        call("sdhip_affine_act", ptr(gv), ldg, ptr(g_slab), Ct, None, 0, None, None, npix, Ct, 1, 0, dt, st)
        dS = gS_in.clone() if gS_in is not None else torch.zeros((groups, 2, Ct), dtype=torch.float64, device=dev)
        grads = []
        fuse = training and growth == 32
        gcount = ops.parallel.global_count(count)     # sync-BN: the replica sums below are all-reduced in place before they are consumed
        pscale = ops.parallel.param_scale()
        pend = None    # norm1 backward sums of the layer processed last, finalized together with this layer's stats_fix
        for li in range(len(layers) - 1, -1, -1):
            layer = layers[li]
            y1, sc1, sh1, mu1, iv1, sc2, sh2, mu2, iv2 = ctx.saved[li]
            Cin = C0 + li * growth
            sl_g = g_slab[:, Cin:Cin + growth]
            sl_x = slab[:, Cin:Cin + growth]
            # (1) total gradient of this layer's 32 output channels (all consumers are already accumulated)
            dy2 = ops.empty_nhwc(B, growth, H, W, dtype, dev)
            if pend is not None:
                # ... in the same launch as the per-channel finalize of the layer above (sdhip_stats_fix_fin)
                dsc, dsh, nl, dgam, dbet, direct, Cf = pend
                call("sdhip_stats_fix_fin", ptr(sl_g), Ct, ptr(sl_x), Ct, ptr(dy2), growth, npix, ptr(dS), Ct, Cin, ptr(dsc), ptr(dsh), NREP,
                     ptr(nl[0]), ptr(nl[1]), ptr(nl[2]), ptr(dgam), ptr(dbet), int(direct), pscale, Cf, groups, float(gcount), dt, st)
                pend = None
            else:
                call("sdhip_stats_fix", ptr(sl_g), Ct, ptr(sl_x), Ct, ptr(dy2), growth, ptr(dS[:, :, Cin:Cin + growth]), Ct,
                     npix, growth, groups, dt, st)   # dS is all zero in eval mode
            # (2) conv2 (3x3): data gradient w.r.t. relu(norm2(y1)), weight gradient
            wd2 = ops.packed_weight(layer.conv2.weight, 'conv', 'dgrad', dtype)
            gp2 = ops.empty_nhwc(B, mid, H, W, dtype, dev)
            fuse2 = training and ops._fused_bn() and dtype == torch.bfloat16 and mid % 8 == 0 and not _lib_mod.DIAG_NO_BNBWD_EPILOGUE
            if fuse2:
                # ... whose epilogue also takes norm2's two backward reductions (no pass of its own over gp2 and y1)
                sums2 = ops._zeros((NREP, groups, 2, mid), torch.float64, dev)[0]
                call("sdhip_conv2d_fwd_bnbwd", ptr(dy2), ptr(wd2), ptr(gp2), ptr(sums2), mid, NREP, ptr(y1), mid,
                     ptr(sc2), ptr(sh2), None, 0, B, H, W, growth, growth, H, W, mid, mid, 3, 3, 1, 1, 1, groups, 0, dt, st)
            else:
                ops._conv_launch(dy2, growth, wd2, gp2, mid, None, None, None, None, B, H, W, growth, H, W, mid, 3, 3, 1, 1, 1, 1,
                                 False, 1, 0, False)
            gw2 = _wgrad(y1, mid, dy2, growth, layer.conv2.weight, B, H, W, mid, growth, 3, 1, sc2, sh2, groups, dt)
            # (3) through relu + norm2's affine, (4) norm2's statistics, (5) into y1
            if fuse2:
                tg, tb = ops._grad_target(layer.norm2.weight), ops._grad_target(layer.norm2.bias)
                direct2 = tg is not None and tb is not None
                dg2 = tg if direct2 else torch.empty(mid, dtype=torch.float32, device=dev)
                db2 = tb if direct2 else torch.empty(mid, dtype=torch.float32, device=dev)
                ops.parallel.all_reduce_sum_(sums2)
                call("sdhip_bn_bwd_apply_fin_d", ptr(gp2), mid, ptr(y1), mid, ptr(gp2), mid, ptr(sc2), ptr(sh2), ptr(sums2), NREP,
                     ptr(layer.norm2.weight), ptr(mu2), ptr(iv2), ptr(dg2), ptr(db2), int(direct2), pscale, npix, mid, groups, float(gcount),
                     1, dt, st)
                if direct2:
                    dg2 = db2 = None
            elif training and ops._fused_bn():
                dg2, db2 = ops.bn_backward_two_phase(gp2, mid, y1, mid, gp2, mid, sc2, sh2, mu2, iv2, layer.norm2.weight,
                                                     layer.norm2.bias, npix, mid, groups, 1, gcount, dt)   # in place: elementwise
            else:
                dg2, db2, dS2 = ops._bn_backward(gp2, mid, y1, mid, gp2, mid, sc2, sh2, mu2, iv2, layer.norm2.weight, npix, mid,
                                                 groups, 1, count, training, dt, beta=layer.norm2.bias)
                if training:
                    call("sdhip_stats_fix", ptr(gp2), mid, ptr(y1), mid, ptr(gp2), mid, ptr(dS2), mid, npix, mid, groups, dt, st)
            # (6) conv1 (1x1): data gradient w.r.t. relu(norm1(slab[:Cin])), weight gradient
            wd1 = ops.packed_weight(layer.conv1.weight, 'conv', 'dgrad', dtype)
            # (7) through relu + norm1's affine, accumulated into the slab gradient; (8) norm1's statistics -> dS
            fused1 = False
            if fuse and li > 0:
                # reductions now; the per-channel finalize rides on the next layer's stats_fix launch
                both, pz = ops._zeros((2, NREP, groups, Cin), torch.float32, dev)
                if dtype == torch.bfloat16 and npix <= FUSE1_MAX_PIX and not _lib_mod.DIAG_NO_BNBWD_EPILOGUE:
                    # small maps: the 1x1 data gradient itself masks, scales and accumulates into the slab gradient and takes
                    # the two reductions (mode 1 of sdhip_conv2d_fwd_bnbwd): gp1 is never written, one launch less per layer
                    rc = _lib_mod._lib.sdhip_conv2d_fwd_bnbwd(ptr(gp2), ptr(wd1), ptr(g_slab), ptr(both), Cin, NREP, ptr(slab), Ct,
                                                              ptr(sc1), ptr(sh1), ptr(g_slab), Ct, B, H, W, mid, mid, H, W, Cin, Ct,
                                                              1, 1, 1, 0, 0, groups, 1, dt, st)
                    if rc == 0:
                        fused1 = True
                    elif rc != _lib_mod.ERR_UNSUPPORTED:
                        raise _lib_mod.SdhipError("sdhip_conv2d_fwd_bnbwd failed (%d): %s" % (rc, _lib_mod._lib.sdhip_last_error().decode()))
            if not fused1:
                gp1 = ops.empty_nhwc(B, Cin, H, W, dtype, dev)
                ops._conv_launch(gp2, mid, wd1, gp1, Cin, None, None, None, None, B, H, W, mid, H, W, Cin, 1, 1, 1, 1, 0, 0,
                                 False, 1, 0, False)
            gw1 = _wgrad(slab, Ct, gp2, mid, layer.conv1.weight, B, H, W, Cin, mid, 1, 0, sc1, sh1, groups, dt)
            if fuse and li > 0:
                if not fused1:
                    call("sdhip_affine_act_bwd", ptr(gp1), Cin, ptr(slab), Ct, ptr(g_slab), Ct, ptr(sc1), ptr(sh1), ptr(both[0]), ptr(both[1]),
                         NREP, npix, Cin, groups, 1, 1, int(pz), dt, st)
                ops.parallel.all_reduce_sum_(both)        # norm1's two reductions [2][NREP][groups][Cin]: global before the finalize
                tg, tb = ops._grad_target(layer.norm1.weight), ops._grad_target(layer.norm1.bias)
                direct = tg is not None and tb is not None
                dgam = tg if direct else torch.empty(Cin, dtype=torch.float32, device=dev)
                dbet = tb if direct else torch.empty(Cin, dtype=torch.float32, device=dev)
                pend = (both[0], both[1], (layer.norm1.weight, mu1, iv1), dgam, dbet, direct, Cin)
                dg1, db1 = (None, None) if direct else (dgam, dbet)
            else:
                dg1, db1, _ = ops._bn_backward(gp1, Cin, slab, Ct, g_slab, Ct, sc1, sh1, mu1, iv1, layer.norm1.weight, npix, Cin,
                                               groups, 1, count, training, dt, accumulate_gx=True, dstats=dS[:, :, :Cin],
                                               accumulate_dstats=True, beta=layer.norm1.bias)
            grads.append((dg1, db1, gw1, dg2, db2, gw2))
        gx0 = ops.empty_nhwc(B, C0, H, W, dtype, dev)
        call("sdhip_stats_fix", ptr(g_slab), Ct, ptr(slab), Ct, ptr(gx0), C0, ptr(dS), Ct, npix, C0, groups, dt, st)
        flat = []
        for g6 in reversed(grads):
            flat.extend(g6)
        return (gx0, None, None) + tuple(flat)


def _stem_s2d_weight(w):
    """The 7x7 / stride 2 / pad 3 stem kernel as the equivalent 4x4 / stride 1 kernel over the 2x2 space-to-depth image:
    out[o] = sum_k w[k] x[2o + k - 3]; with k + 1 = 2a + r (a < 4, r < 2; the slot k = -1 is a zero) this is
    sum_a sum_r w'[r][a] X[r][o + a - 2], X[r][j] = x[2j + r] — 16 taps over 4*C channels instead of 49 taps over C = 3
    channels that fill 3 of the 32 lanes of an MFMA k-step.  Plain (differentiable) tensor ops on a 9 K-element
    weight: autograd carries the gradient of the 4x4 form back to the (Cout, C, 7, 7) parameter."""
    Cout, C = w.shape[0], w.shape[1]
    wp = torch.nn.functional.pad(w, (1, 0, 1, 0))
    return wp.reshape(Cout, C, 4, 2, 4, 2).permute(0, 3, 5, 1, 2, 4).reshape(Cout, 4 * C, 4, 4)


def _space_to_depth2(x):
    """(B, C, H, W) -> (B, 4C, H/2, W/2), channel (ry*2 + rx)*C + c = x[b, c, 2h + ry, 2w + rx]; one strided copy into a
    channels-last buffer whose pixel stride is padded to 16 bytes."""
    B, C, H, W = x.shape
    y, _ = ops.alloc_nhwc(B, 4 * C, H // 2, W // 2, x.dtype, x.device)
    y.unflatten(1, (2, 2, C)).copy_(x.reshape(B, C, H // 2, 2, W // 2, 2).permute(0, 3, 5, 1, 2, 4))
    return y


class _Stem(torch.autograd.Function):
    """conv0 (7x7 / 2) -> tap 0 (raw) and relu(norm0(.)): both outputs are used (models/densenet.py:222-225).
    A (Cout, 4C, 4, 4) weight selects the space-to-depth form (x is then the space-to-depth image)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, bn, groups):
        # an output nobody consumes (tap 0 in the configurations without the full-resolution pyramid) must not come back as a
        # materialised NCHW zero tensor: that is a fill, a layout conversion and an add over 67 MB for nothing
        ctx.set_materialize_grads(False)
        B, Cin, H, W = x.shape
        xv, ldx = ops.nhwc_view(x)
        if weight.shape[-1] == 4:
            spec = ops.ConvSpec('conv', 4, 4, 1, 1, 2, 2, H, W)
        else:
            spec = ops.conv_spec(x, weight, 'conv', 2, 1, 3)
        Cout = weight.shape[0]
        wp = ops.packed_weight(weight, 'conv', 'fwd', x.dtype)
        c0 = ops.empty_nhwc(B, Cout, spec.Ho, spec.Wo, x.dtype, x.device)
        train = bn.training
        ws = ops._zeros((NREP, groups, 2, Cout), torch.float64, x.device)[0] if train else None
        ops._conv_launch(xv, ldx, wp, c0, Cout, None, None, None, ws, B, H, W, Cin, spec.Ho, spec.Wo, Cout, spec.kh, spec.kw,
                         spec.stride, spec.dil, spec.pad_t, spec.pad_l, False, groups, 0, False, NREP)
        count = (B // groups) * spec.Ho * spec.Wo
        scale, shift, mean, invstd = ops._bn_finalize(ws, NREP, bn, count, groups)
        f = ops.empty_nhwc(B, Cout, spec.Ho, spec.Wo, x.dtype, x.device)
        call("sdhip_affine_act", ptr(c0), Cout, ptr(f), Cout, None, 0, ptr(scale), ptr(shift), B * spec.Ho * spec.Wo, Cout,
             groups, 1, dtype_code(x), stream_ptr())
        ctx.cfg = (spec, groups, ldx, count, train)
        ctx.save_for_backward(xv, weight, gamma, beta, c0, scale, shift, mean, invstd)
        return c0, f

    @staticmethod
    def backward(ctx, g0, gf):
        xv, weight, gamma, beta, c0, scale, shift, mean, invstd = ctx.saved_tensors
        spec, groups, ldx, count, train = ctx.cfg
        B, Cout = c0.shape[0], c0.shape[1]
        npix = B * spec.Ho * spec.Wo
        dt = dtype_code(xv)
        # gradient w.r.t. the raw conv output = tap-0 consumers + the norm0/relu branch
        if g0 is None and gf is None:
            return None, None, None, None, None, None
        graw = ops.empty_nhwc(B, Cout, spec.Ho, spec.Wo, xv.dtype, xv.device)
        if g0 is not None:
            gv, ldg = ops.nhwc_view(g0)
            call("sdhip_affine_act", ptr(gv), ldg, ptr(graw), Cout, None, 0, None, None, npix, Cout, 1, 0, dt, stream_ptr())
        dgamma = dbeta = None
        if gf is not None:
            fv, ldf = ops.nhwc_view(gf)
            dgamma, dbeta, dS = ops._bn_backward(fv, ldf, c0, Cout, graw, Cout, scale, shift, mean, invstd, gamma, npix, Cout,
                                                 groups, 1, count, train, dt, accumulate_gx=g0 is not None, beta=beta)
            if train:
                call("sdhip_stats_fix", ptr(graw), Cout, ptr(c0), Cout, ptr(graw), Cout, ptr(dS), Cout, npix, Cout, groups, dt,
                     stream_ptr())
        _, gw, _ = ops._conv_backward(spec, xv, ldx, weight, graw, Cout, None, None, False, 1, False, True)
        return None, gw, dgamma, dbeta, None, None


class _DenseLayer(nn.Module):
    """models/densenet.py:25-93 — parameter container; the block drives the kernels."""

    def __init__(self, cin, growth, bn_size):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(cin)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(cin, bn_size * growth, 1, bias=False)
        self.norm2 = nn.BatchNorm2d(bn_size * growth)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(bn_size * growth, growth, 3, padding=1, bias=False)


class _DenseBlock(nn.ModuleDict):
    """models/densenet.py:96-116."""

    def __init__(self, n, cin, bn_size, growth):
        super().__init__()
        for i in range(n):
            self['denselayer%d' % (i + 1)] = _DenseLayer(cin + i * growth, growth, bn_size)

    def forward(self, x, groups=1):
        params = []
        for layer in self.values():
            params += [layer.norm1.weight, layer.norm1.bias, layer.conv1.weight,
                       layer.norm2.weight, layer.norm2.bias, layer.conv2.weight]
        return _DenseBlockFn.apply(x, self, groups, *params)   # (slab, statistics)


class _Transition(nn.Sequential):
    """models/densenet.py:119-128: norm + relu fused into the 1x1 conv's prologue (pool stays outside)."""

    def __init__(self, cin, cout):
        super().__init__(OrderedDict(norm=nn.BatchNorm2d(cin), relu=nn.ReLU(inplace=True),
                                     conv=nn.Conv2d(cin, cout, 1, bias=False)))

    def forward(self, slab, stats, groups=1):
        return ops.bn_conv(slab, stats, self.norm, self.conv.weight, padding=0, groups=groups)


class DenseNet(nn.Module):
    """models/densenet.py:131-245: returns the five taps; tap 0 is the raw conv0 output, taps 1-3 the transition
    outputs before the 2x2 average pool, tap 4 relu(norm5(.))."""

    def __init__(self, growth_rate=32, block_config=(6, 12, 24, 16), num_init_features=64, bn_size=4, num_classes=1000):
        super().__init__()
        self.conv0 = nn.Conv2d(3, num_init_features, 7, stride=2, padding=3, bias=False)
        self.features = nn.Sequential(OrderedDict(norm0=nn.BatchNorm2d(num_init_features), relu0=nn.ReLU(inplace=True),
                                                  pool0=nn.MaxPool2d(3, stride=2, padding=1)))
        blocks, c = [], num_init_features
        for i, n in enumerate(block_config):
            blocks.append(_DenseBlock(n, c, bn_size, growth_rate))
            c += n * growth_rate
            if i != len(block_config) - 1:
                blocks.append(_Transition(c, c // 2))
                c //= 2
        self.denseblock = nn.ModuleList(blocks)
        self.norm5 = nn.BatchNorm2d(c)
        self.classifier = nn.Linear(c, num_classes)   # never called by the reference either (kept for the state_dict)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.constant_(m.bias, 0)

    def forward(self, x, groups=1):
        w0 = self.conv0.weight
        # The space-to-depth form sums the 147 products of an output in another order than the 7x7 form.  Exact in exact
        # arithmetic and at the f32 rounding level per output (2e-7), but train-mode BatchNorm over tiny batches amplifies
        # any such reordering (tests/diag/gpu_stem_diag.py), so the f32 parity path keeps the 7x7 form that the golden vectors
        # were captured with; the bf16 throughput path takes the 2 % faster step.  SDHIP_STEM_S2D=0/1 forces either.
        s2d = _lib_mod.DIAG_STEM_S2D
        use_s2d = (x.dtype == torch.bfloat16) if s2d is None else s2d == "1"
        if use_s2d and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0:
            # (the image may arrive zero-padded to 8 channels for the 7x7 kernels: only the weight's channels are real)
            x, w0 = _space_to_depth2(x[:, :w0.shape[1]]), _stem_s2d_weight(w0)
        c0, f = _Stem.apply(x, w0, self.features.norm0.weight, self.features.norm0.bias, self.features.norm0, groups)
        taps = [c0]
        f = ops.maxpool3s2(f)
        stats = None
        for i, blk in enumerate(self.denseblock):
            if i % 2 == 0:
                f, stats = blk(f, groups)
            else:
                f = blk(f, stats, groups)
                taps.append(f)
                f = ops.avgpool(f, 2)
        taps.append(ops.bn_act(f, stats, self.norm5, act=1, groups=groups))
        return taps


def densenet121(pretrained=False, progress=True, **kwargs):
    if pretrained:
        raise RuntimeError("pretrained=True downloads ImageNet weights (models/densenet.py:256); load a state_dict instead")
    return DenseNet(32, (6, 12, 24, 16), 64, **kwargs)
