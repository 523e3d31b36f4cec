"""Poisoned-allocation bisect of the HANet head: which op reads uninitialised memory?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
_empty, _empty_like = torch.empty, torch.empty_like
def _poison(t):
    if t.is_cuda and t.is_floating_point():
        t.fill_(float("nan"))
    return t
torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))
from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N, ops
from pmt_learning_for_semantic_segmentation_and_disparity_amd.hanet import HANet_Conv
import torch.nn as nn

def fin(t):
    return bool(torch.isfinite(t.float()).all())

for dt in (torch.float32, torch.bfloat16):
    print("==", dt, flush=True)
    # (1) conv + BN + ReLU with one output channel on (B,64,64,1) row descriptors, then a consumer with 1 -> 2 channels
    for cin, cout in ((64, 1), (1, 2), (2, 19), (64, 8)):
        torch.manual_seed(0)
        x = torch.randn(2, cin, 64, 1, device="cuda").to(dt)
        xx, ldx = ops.alloc_nhwc(2, cin, 64, 1, dt, "cuda")
        xx.copy_(x); xx.requires_grad_(True)
        w = (torch.randn(cout, cin, 1, 1, device="cuda") * 0.3).requires_grad_(True)
        bn = nn.BatchNorm2d(cout).cuda().train()
        y = ops.conv_bn_act(xx, w, bn, padding=(0, 0), act=1)
        gy, _ = ops.alloc_nhwc(2, cout, 64, 1, dt, "cuda")
        gy.copy_(torch.randn(2, cout, 64, 1, device="cuda").to(dt))
        y.backward(gy)
        print("conv_bn_act %d->%d k1: y %s  gx %s  gw %s  dgamma %s dbeta %s" % (cin, cout, fin(y), fin(xx.grad), fin(w.grad), fin(bn.weight.grad), fin(bn.bias.grad)), flush=True)
        w3 = (torch.randn(cout, cin, 3, 1, device="cuda") * 0.3).requires_grad_(True)
        xx.grad = None
        y = ops.conv_bn_act(xx, w3, bn, padding=(1, 0), act=1)
        y.backward(gy)
        print("conv_bn_act %d->%d k3x1: y %s  gx %s  gw %s" % (cin, cout, fin(y), fin(xx.grad), fin(w3.grad)), flush=True)
        b = torch.zeros(cout, device="cuda", requires_grad=True)
        xx.grad = None
        y = ops.conv2d(xx, w3, b, padding=(1, 0), act=2)
        y.backward(gy)
        print("conv2d+sigmoid %d->%d k3x1: y %s  gx %s  gw %s gb %s" % (cin, cout, fin(y), fin(xx.grad), fin(w3.grad), fin(b.grad)), flush=True)
    # (2) the module
    torch.manual_seed(0)
    m = HANet_Conv(64, 19, pooling='max', pos_rfactor=2, dropout_prob=0.1).cuda().train()
    x = torch.randn(2, 64, 128, 128, device="cuda").to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = torch.randn(2, 19, 256, 256, device="cuda").to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    H, W = 256, 256
    h = ((torch.arange(0, H) * 1024 // H).unsqueeze(0).unsqueeze(2).expand(2, -1, W) // 8).cuda()
    w_ = ((torch.arange(0, W) * 2048 // W).unsqueeze(0).unsqueeze(1).expand(2, H, -1) // 16).cuda()
    y, logits = m(x, out, (h, w_), attention_loss=True)
    y.float().pow(2).mean().backward()
    print("HANet module: y %s gx %s gout %s" % (fin(y), fin(x.grad), fin(out.grad)), {k: fin(p.grad) for k, p in m.named_parameters() if p.grad is not None}, flush=True)
