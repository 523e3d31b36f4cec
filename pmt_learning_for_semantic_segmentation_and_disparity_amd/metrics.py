"""On-device step metrics.

The reference computes its per-step scores on the host: `lossSeg_fn` / `lossDisp_fn`
(losses/multiLosses.py:8-125,131-157) copy the segmentation logits, the log-softmax, the
one-hot target and both disparity maps to numpy and run `SegAccuracyNp`, `GetSegMetricsNp`,
`unnormalizedErrorNP` and `GetDispMetricsNp` (util/utilTorchLoss.py:221-236,251-303,363-370,
318-343) — a device synchronisation and ~40 MB of PCIe traffic per step at B=8, 256x512,
plus six `cv2.imwrite` JPEG dumps that are not reproduced here.

`StepMetrics.update()` is one kernel launch (`sdhip_step_metrics`) that accumulates integer
counters and f64 sums in device memory; nothing is copied until `compute()` is called, so a
training loop can report once per epoch (or every N steps) with a single ~1 KB transfer.
The returned names and definitions are the reference's, quirks included (see `compute`).
"""
import math

import torch

from . import _lib
from ._lib import call, dtype_code, ptr, stream_ptr
from .ops import _require_gpu, nhwc_view

N_COUNTS, N_SUMS, SUM_STRIDE = 10, 4, 32      # SDHIP_METRIC_COUNTS / _SUMS / _SUM_STRIDE of include/sdhip.h
NREP = 32                                    # replicas the workgroups spread their closing atomics over


def _safe_div(a, b):
    """sklearn's zero_division="warn" convention: 0 when the denominator is 0."""
    return a / b if b else 0.0


class StepMetrics:
    """Accumulator over the steps of one reporting interval.  All state lives in device memory and `update` is a
    single kernel launch, so it can sit inside a captured training step (train.TrainStep(metrics=...)).

    labels     number of classes L of the segmentation head (`labels = seg.shape[1]`, multiLosses.py:23)
    max_disp   the `max_disp` the reference scales the >3 px test with (multiLosses.py:150)
    mask_invalid  True for the datasets whose disparity target has holes (`zeros = (disp > 0)`, multiLosses.py:134-139)
    """

    def __init__(self, labels, max_disp=1.0, mask_invalid=False, device="cuda"):
        if not torch.cuda.is_available():
            raise _lib.SdhipError("StepMetrics needs a GPU; there is no CPU path")
        self.labels, self.max_disp, self.mask_invalid = int(labels), float(max_disp), bool(mask_invalid)
        self.rep_stride = ((self.labels ** 2 + N_COUNTS + 31) // 32) * 32       # whole 256-byte lines per replica
        self.counts = torch.zeros(NREP, self.rep_stride, dtype=torch.int64, device=device)
        self.sums = torch.zeros(NREP, SUM_STRIDE, dtype=torch.float64, device=device)

    def reset(self):
        self.counts.zero_()
        self.sums.zero_()

    def update(self, seg_pred=None, seg_full=None, disp_pred=None, disp=None):
        """seg_pred: raw logits (B,L,H,W) of the head being scored (`init_pred_np`; its argmax equals the argmax of
        the log-softmax the reference hands to SegAccuracyNp); seg_full: one-hot f32 (B,L or L+1,H,W);
        disp_pred: (B,1,H,W); disp: f32 (B,1,H,W).  Either pair may be omitted.  Asynchronous."""
        _require_gpu(seg_pred, seg_full, disp_pred, disp)
        if seg_pred is None and disp_pred is None:
            raise _lib.SdhipError("StepMetrics.update: nothing to score")
        ref = seg_pred if seg_pred is not None else disp_pred
        B, H, W = ref.shape[0], ref.shape[2], ref.shape[3]
        dt = dtype_code(ref)
        sv = tv = dv = gv = None
        lds = ldt = Ct = 0
        if seg_pred is not None:
            if seg_full is None:
                raise _lib.SdhipError("StepMetrics.update: seg_full is required with seg_pred")
            if seg_pred.shape[1] != self.labels:
                raise _lib.SdhipError("StepMetrics.update: %d logits per pixel, labels=%d" % (seg_pred.shape[1], self.labels))
            sv, lds = nhwc_view(seg_pred)
        if seg_full is not None:
            if seg_full.dtype != torch.float32 or seg_full.shape[0] != B or tuple(seg_full.shape[2:]) != (H, W):
                raise _lib.SdhipError("StepMetrics.update: seg_full must be f32 (B,C,H,W) matching the prediction")
            tv, ldt = nhwc_view(seg_full)
            Ct = seg_full.shape[1]
        if disp_pred is not None:
            if disp is None or disp.dtype != torch.float32 or disp.numel() != B * H * W or disp_pred.numel() != B * H * W:
                raise _lib.SdhipError("StepMetrics.update: disp must be f32 with one value per pixel of disp_pred")
            if disp_pred.dtype != ref.dtype:
                raise _lib.SdhipError("StepMetrics.update: seg_pred and disp_pred must share a dtype")
            dv, ldd = nhwc_view(disp_pred)
            if ldd != 1:
                dv = disp_pred.contiguous()
            gv = disp.contiguous()
        call("sdhip_step_metrics", ptr(sv) if sv is not None else None, lds, ptr(tv) if tv is not None else None, ldt, Ct,
             ptr(dv) if dv is not None else None, ptr(gv) if gv is not None else None, ptr(self.counts), ptr(self.sums),
             NREP, self.rep_stride, B, H * W, self.labels, self.max_disp, int(self.mask_invalid), dt, stream_ptr())
        self._keep = (sv, tv, dv, gv)          # alive until the next call (the launch is asynchronous)

    def totals(self):
        """(counts[L*L + N_COUNTS] int64, sums[N_SUMS] f64) on the device, replicas summed."""
        return self.counts.sum(0)[:self.labels ** 2 + N_COUNTS], self.sums.sum(0)[:N_SUMS]

    def compute(self):
        """One device->host copy; returns a dict with the reference's names.

        pixelAcc, conf_matrix                        SegAccuracyNp (utilTorchLoss.py:221-236)
        pixelPrec, pixelRecall, pixelF1, pixelBF1    GetSegMetricsNp (:251-303): image 0, channel 1, sklearn "micro"
        err, val_pxl                                 unnormalizedErrorNP (:363-370), summed over the interval
        dispRMSE, dispSqRel, BdispSqRel              GetDispMetricsNp (:318-343): image 0 of each step, pooled
        BdispRMSE                                    as the reference computes it: sqrt(dispRMSE) (:339-340 take the
                                                     square root of the already reduced scalar); the masked value the
                                                     name promises is reported as BdispRMSE_masked
        """
        L = self.labels
        c = self.totals()[0].cpu().numpy()
        s = self.sums.sum(0)[:N_SUMS].cpu().numpy()
        conf = c[:L * L].reshape(L, L).copy()
        k = c[L * L:]
        tp, fp, fn = int(k[0]), int(k[1]), int(k[2])
        prec, rec = _safe_div(tp, tp + fp), _safe_div(tp, tp + fn)
        n1 = int(k[9])       # counted on the device, so launches replayed from a hipGraph are included
        rmse = math.sqrt(s[0] / n1) if n1 else float("nan")
        nb = int(k[8])
        return {
            "pixelAcc": _safe_div(float(conf.trace()), float(conf.sum())) if conf.sum() else float("nan"),
            "conf_matrix": conf,
            "pixelPrec": prec, "pixelRecall": rec, "pixelF1": _safe_div(2 * prec * rec, prec + rec),
            "pixelBF1": _safe_div(int(k[4]), int(k[5])),
            "err": float(k[6]), "val_pxl": float(k[7]),
            "dispRMSE": rmse, "dispSqRel": (s[1] / n1) if n1 else float("nan"),
            "BdispRMSE": math.sqrt(rmse) if n1 else float("nan"),
            "BdispRMSE_masked": math.sqrt(s[2] / nb) if nb else float("nan"),
            "BdispSqRel": (s[3] / nb) if nb else float("nan"),
        }
