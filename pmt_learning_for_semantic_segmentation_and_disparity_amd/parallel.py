"""Data-parallel exchange steps of the hot path (one process per GPU, torch.distributed; backend 'nccl' is RCCL
over xGMI on ROCm).  The path shards over stereo pairs; only two things cross ranks:

  1. the flat f32 gradient buffer: ONE all-reduce(SUM) per step (train.TrainStep), Adam divides by world size;
  2. BatchNorm batch statistics (the reference converts to nn.SyncBatchNorm, torch_implementation.py:739):
     forward  — all-reduce(SUM) of the f64 (sum x, sum x^2) per channel, count scaled by world size;
     backward — all-reduce(SUM) of (dscale, dshift) = (sum gy*act'*x, sum gy*act') before the statistics gradient
                is formed.
     The exchange is IN PLACE on the producer's replica buffer ([replica][group][2][C], the buffer the conv epilogue's
     atomics were spread over): summing replica-wise over ranks and folding the replicas afterwards is the same sum, and
     the kernels that consume the buffer — the fused finalize + normalise passes, the consumer-side finalize of the DenseNet
     convolutions, the fused backward passes — are exactly the single-GPU ones.  A multi-rank step is the single-rank step
     plus one small collective per BatchNorm and direction; dgamma / dbeta come out of the global sums scaled by 1 / world
     (param_scale), which the gradient all-reduce (SUM) turns back into the total.
     Semantics = torch.nn.SyncBatchNorm = sync_batchnorm/batchnorm.py:114-126 (count-weighted mean, biased variance
     for normalisation, unbiased for running_var).

These helpers take any tensor (CPU tensors with gloo in the tests, GPU tensors with RCCL in production).
"""
import torch

_state = {"pg": None, "world": 1}


def configure(process_group=None, world_size=1):
    _state["pg"], _state["world"] = process_group, int(world_size)


def world_size():
    return _state["world"]


def all_reduce_sum_(t):
    """In-place sum over ranks (no-op for a single rank)."""
    if _state["world"] > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM, group=_state["pg"])
    return t


def global_count(local_count):
    return local_count * _state["world"]


def param_scale():
    """Factor for BatchNorm parameter gradients formed from sums that are ALREADY global (the fused backward kernels read the
    all-reduced replica sums): each rank contributes 1 / world of them, the flat gradient all-reduce (SUM) restores the total."""
    return 1.0 / _state["world"]


def bn_scale_shift_from_sums(s1, s2, count, gamma, beta, eps):
    """Reference maths of the finalize kernel on (already all-reduced) sums — used by the CPU tests to show that
    the exchanged quantities reproduce single-process BatchNorm on the concatenated batch."""
    mean = s1 / count
    var = (s2 / count - mean * mean).clamp_min(0)
    invstd = (var + eps).rsqrt()
    scale = gamma.double() * invstd
    return scale, beta.double() - mean * scale, mean, var
