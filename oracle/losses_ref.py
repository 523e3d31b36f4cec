"""ORACLE (test infrastructure): plain-torch restatement of the Lovasz-softmax term of the training loss
(util/lovasz_losses.py:153-199 with classes='present', as called from losses/multiLosses.py:70-72)."""
import torch


def lovasz_softmax_onehot(logits, labels_onehot, ignore_void=False):
    """Lovasz-softmax (util/lovasz_losses.py:153-199, classes='present') in plain torch ops — the checker of the
    native kernel (sdhip_lovasz_softmax) in the tests and part of bench.py's CPU baseline step."""
    B, C, H, W = logits.shape
    p = torch.softmax(logits.float(), 1).permute(0, 2, 3, 1).reshape(-1, C)
    fg_all = labels_onehot.permute(0, 2, 3, 1).reshape(-1, C)
    # void pixels (no positive entry in the one-hot row: `ignore=19` after the 20th channel is dropped,
    # losses/multiLosses.py:19-21) are removed before anything else (flatten_probas, util/lovasz_losses.py:202-216)
    # ignore_void=False is `ignore=None` (roses / garden, losses/multiLosses.py:11-17): labels = argmax(one-hot), so an
    # all-zero row is class 0 and stays in
    if ignore_void:
        valid = fg_all.max(1).values > 0
        p, fg_all = p[valid], fg_all[valid]
    else:
        fg_all = torch.nn.functional.one_hot(fg_all.argmax(1), C).to(p.dtype)
    if p.numel() == 0:
        return logits.sum() * 0.0
    total = p.new_zeros(())
    present = p.new_zeros(())
    for c in range(C):
        fg = fg_all[:, c]
        err = (fg - p[:, c]).abs()
        err_s, perm = torch.sort(err, 0, descending=True)
        fg_s = fg[perm]
        gts = fg_s.sum()
        inter = gts - fg_s.cumsum(0)
        union = gts + (1.0 - fg_s).cumsum(0)
        jac = 1.0 - inter / union
        jac = torch.cat([jac[:1], jac[1:] - jac[:-1]])
        has = (gts > 0).to(p.dtype)
        total = total + has * torch.dot(err_s, jac.detach())
        present = present + has
    return total / present.clamp_min(1.0)




def train_loss_ref(seg1, disp, seg2, seg_t, disp_t, use_lovasz=True, mask_invalid_disp=False, ignore_void=False):
    """The loss of the timed step: CE(seg1) + CE(seg2) [+ Lovasz(seg2)] + L1(disp)
    (torch_implementation.py:279,293,304,325; losses/multiLosses.py:66-72,134-141; util/utilTorchLoss.py:373-378)."""
    import torch.nn.functional as F
    ce = lambda y: torch.mean(torch.sum(-seg_t * F.log_softmax(y.float(), 1), 1))
    z = (disp_t > 0).float() if mask_invalid_disp else 1.0
    loss = ce(seg1) + ce(seg2) + F.l1_loss(disp.float() * z, disp_t * z)
    if use_lovasz:
        loss = loss + lovasz_softmax_onehot(seg2, seg_t, ignore_void)
    return loss
