"""ORACLE (test infrastructure, never shipped or timed as the product): numpy restatement of the reference's
host-side step metrics.  Pinned by tests/golden/metrics.npz, which oracle/make_golden.py produced by calling the
reference's own functions (util/utilTorchLoss.py) on the same inputs.

Each function follows the cited lines; the sklearn scores are written out as the count ratios sklearn's
average="micro" reduces to, so this file needs numpy only.
"""
import numpy as np


def seg_accuracy(outputs, gt, labels):
    """SegAccuracyNp, util/utilTorchLoss.py:221-236 -> (acc, conf_matrix)."""
    gt_seg = gt.argmax(1)
    pred_seg = outputs.argmax(1)
    mask = gt_seg != labels
    acc = (pred_seg == gt_seg)[mask].mean()
    count = np.bincount(labels * gt_seg[mask] + pred_seg[mask], minlength=labels ** 2)
    return acc, count.reshape(labels, labels)


def seg_metrics(outputs, gt):
    """GetSegMetricsNp, util/utilTorchLoss.py:251-303 (without the cv2.imwrite dumps) -> precision, recall, f1, Bf1.
    Image 0, channel 1 only.  precision/recall/f1 are sklearn scores of two (H, W) 0/1 arrays, i.e. multilabel
    "micro" = pooled TP/FP/FN; Bf1 is a "micro" f1 of two 1-D selections, i.e. the fraction of equal values."""
    gt_img = gt[0][1]
    raw = outputs[0][1]
    branch = np.logical_or(gt_img == 1.0, raw == 1.0)          # :257, taken before the thresholding below
    pred = raw.copy()
    pred[pred > 0] = 1                                          # :265
    pred[pred < 0] = 0                                          # :266
    tp = np.sum((pred == 1) & (gt_img != 0))
    fp = np.sum((pred == 1) & (gt_img == 0))
    fn = np.sum((pred != 1) & (gt_img != 0))
    div = lambda a, b: a / b if b else 0.0
    precision, recall = div(tp, tp + fp), div(tp, tp + fn)
    f1 = div(2 * precision * recall, precision + recall)
    bf1 = div(np.sum(pred[branch] == gt_img[branch]), int(branch.sum()))
    return precision, recall, f1, bf1


def unnormalized_error(y_, y, max_disp):
    """unnormalizedErrorNP, util/utilTorchLoss.py:363-370 -> (#bad, #valid)."""
    th = (y > 0) * 1.0
    e = np.abs(y_ * max_disp - y * max_disp) * th
    return np.sum((e > 3.0) * 1.0), np.sum(th)


def disp_metrics(outputs, gt, seg_full):
    """GetDispMetricsNp, util/utilTorchLoss.py:318-343 -> dispRMSE, dispSqRel, BdispRMSE, BdispSqRel.
    BdispRMSE is the reference's: the square root of the already reduced dispRMSE (:339-340)."""
    branch = seg_full[0][1] == 1.0
    g, p = gt[0][0], outputs[0][0]
    rmse = np.sqrt(((g - p) ** 2).mean())
    with np.errstate(divide="ignore", invalid="ignore"):
        sqrel = np.mean(((g - p) ** 2) / g)
        bsqrel = np.mean(((g[branch] - p[branch]) ** 2) / g[branch])
    return rmse, sqrel, np.sqrt(rmse.mean()), bsqrel


def step_metrics(seg_pred, seg_full, disp_pred, disp, labels, max_disp, mask_invalid=False):
    """The block of lossSeg_fn / lossDisp_fn (losses/multiLosses.py:116-125,146-154) as one dict (names as there)."""
    out = {}
    if seg_pred is not None:
        acc, conf = seg_accuracy(seg_pred, seg_full, labels)
        p, r, f1, bf1 = seg_metrics(seg_pred, seg_full)
        out.update(pixelAcc=acc, conf_matrix=conf, pixelPrec=p, pixelRecall=r, pixelF1=f1, pixelBF1=bf1)
    if disp_pred is not None:
        zeros = (disp > 0) * np.float32(1.0) if mask_invalid else np.float32(1.0)
        dp, dg = disp_pred * zeros, disp * zeros
        err, val = unnormalized_error(dp, dg, max_disp)
        rmse, sqrel, brmse, bsqrel = disp_metrics(dp, dg, seg_full)
        out.update(err=err, val_pxl=val, dispRMSE=rmse, dispSqRel=sqrel, BdispRMSE=brmse, BdispSqRel=bsqrel)
    return out
