"""ORACLE (test infrastructure): plain-torch restatement of the Lovasz-softmax term of the training loss
(util/lovasz_losses.py:153-199 with classes='present', as called from losses/multiLosses.py:70-72)."""
import torch


def lovasz_softmax_onehot(logits, labels_onehot):
    """Lovasz-softmax (util/lovasz_losses.py:153-199, classes='present') in plain torch ops — the checker of the
    native kernel (sdhip_lovasz_softmax) in the tests and part of bench.py's CPU baseline step."""
    B, C, H, W = logits.shape
    p = torch.softmax(logits.float(), 1).permute(0, 2, 3, 1).reshape(-1, C)
    fg_all = labels_onehot.permute(0, 2, 3, 1).reshape(-1, C)
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


