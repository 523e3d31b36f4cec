"""Height-driven attention head — `HANet_Conv` / `PosEncoding1D` of the reference
(models_hanet/HANet.py:9-128, models_hanet/PosEmbedding.py:7-85), same constructor, parameter names and state_dict keys.

Data-sized steps (row max-pool of the feature map, channel dropout, row-wise re-weighting of the logits) are kernels of
libsdhip (csrc/hanet.hip); the 1-D convolutions / BatchNorm1d / sigmoid / linear resize of the tiny (B, C, L) row
descriptor run through the library's conv / BatchNorm / resize kernels on (B, C, L, 1) images."""
import math
import numpy as np
import torch
import torch.nn as nn

from . import ops


def get_sinusoid_encoding_table(n_position, d_hid):
    """models_hanet/PosEmbedding.py:7-28 (note the reference overrides its own cycle choice: 10 if d_hid > 50 else 100)."""
    cycle = 10 if d_hid > 50 else 100
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    hid = np.arange(d_hid)[None, :]
    table = pos / np.power(cycle, 2 * (hid // 2) / d_hid)
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    return torch.FloatTensor(table)


class PosEncoding1D(nn.Module):
    """models_hanet/PosEmbedding.py:49-85 (pos_noise == 0 only: the hot path never enables the noise)."""

    def __init__(self, pos_rfactor, dim, pos_noise=0.0):
        super().__init__()
        if pos_noise > 0.0:
            raise NotImplementedError("positional noise is not on the hot path")
        self.pos_layer = nn.Embedding.from_pretrained(get_sinusoid_encoding_table((128 // pos_rfactor) + 1, dim) + 1, freeze=True)
        self.pos_rfactor = pos_rfactor

    def encoding(self, L, pos):
        """(B, dim, L, 1) additive encoding: index plumbing on tiny integer tensors (row index // rfactor, nearest resize
        to L, table lookup) exactly as the reference's forward."""
        pos_h, _ = pos                                               # B x H x W
        pos_h = pos_h // self.pos_rfactor
        pos_h = pos_h[:, :, :1].unsqueeze(1).squeeze(3)              # B x 1 x H
        pos_h = nn.functional.interpolate(pos_h.float(), size=L, mode='nearest').long()
        pe = self.pos_layer(pos_h).transpose(1, 3).squeeze(3)        # B x dim x L
        return pe.unsqueeze(3)

    def forward(self, x, pos):
        pe = self.encoding(x.shape[2], pos).to(x.dtype)
        return ops.add(x, pe.contiguous(memory_format=torch.channels_last))


class HANet_Conv(nn.Module):
    _DROP_LAYER = 9001

    def __init__(self, in_channel, out_channel, kernel_size=3, r_factor=64, layer=3, pos_injection=2, is_encoding=1,
                 pos_rfactor=8, pooling='mean', dropout_prob=0.0, pos_noise=0.0):
        super().__init__()
        if pooling != 'max' or is_encoding != 1 or layer not in (2, 3) or pos_injection not in (1, 2):
            raise NotImplementedError("native HANet: max row pooling, sinusoid encoding, 2 or 3 layers (what dsnet_t2.py:1137 builds)")
        self.pooling, self.pos_injection, self.layer, self.dropout_prob = pooling, pos_injection, layer, dropout_prob
        self.sigmoid = nn.Sigmoid()
        if r_factor > 0:
            mid_1 = math.ceil(in_channel / r_factor)
        else:
            mid_1 = in_channel * (-r_factor)
        if dropout_prob > 0:
            self.dropout = nn.Dropout2d(dropout_prob)
        self.attention_first = nn.Sequential(nn.Conv1d(in_channel, mid_1, 1, bias=False), nn.BatchNorm1d(mid_1), nn.ReLU(inplace=True))
        if layer == 2:
            self.attention_second = nn.Sequential(nn.Conv1d(mid_1, out_channel, kernel_size, padding=kernel_size // 2, bias=True))
        else:
            mid_2 = mid_1 * 2
            self.attention_second = nn.Sequential(nn.Conv1d(mid_1, mid_2, 3, padding=1, bias=True), nn.BatchNorm1d(mid_2),
                                                  nn.ReLU(inplace=True))
            self.attention_third = nn.Sequential(nn.Conv1d(mid_2, out_channel, kernel_size, padding=kernel_size // 2, bias=True))
        self.rows = 128 // pos_rfactor
        if pos_rfactor > 0:
            if pos_injection == 1:
                self.pos_emb1d_1st = PosEncoding1D(pos_rfactor, dim=in_channel, pos_noise=pos_noise)
            else:
                self.pos_emb1d_2nd = PosEncoding1D(pos_rfactor, dim=mid_1, pos_noise=pos_noise)
        self._w4 = {}

    def _weight4(self, conv):
        """(Cout, Cin, k, 1) view of a Conv1d weight, created once: the pack cache and the gradient routing key on it."""
        v = self._w4.get(id(conv))
        if v is None or v.data_ptr() != conv.weight.data_ptr():
            v = conv.weight.unsqueeze(-1)
            self._w4[id(conv)] = v
        return v

    def _conv_bn_relu(self, x, conv, bn):
        pad = (conv.padding[0], 0)
        w4 = self._weight4(conv)
        if bn.training:
            # BatchNorm removes the conv bias from its output; only the running mean sees it
            y = ops.conv_bn_act(x, w4, bn, padding=pad, act=1)
            if conv.bias is not None and bn.momentum is not None:
                with torch.no_grad():
                    bn.running_mean.add_(conv.bias.detach().to(bn.running_mean.dtype), alpha=bn.momentum)
            return y
        y = ops.conv2d(x, w4, conv.bias, padding=pad)
        return ops.bn_act(y, None, bn, act=1)

    def forward(self, x, out, pos=None, return_attention=False, return_posmap=False, attention_loss=False):
        if return_posmap:
            raise NotImplementedError("return_posmap is a visualisation path")
        H = out.size(2)
        x1d = ops.rowpool_max(x, self.rows)                                   # (B, C, L, 1)
        if pos is not None and self.pos_injection == 1:
            x1d = self.pos_emb1d_1st(x1d, pos)
        if self.dropout_prob > 0:
            x1d = ops.dropout_channels(x1d, self.dropout_prob, self.training, self._DROP_LAYER)
        x1d = self._conv_bn_relu(x1d, self.attention_first[0], self.attention_first[1])
        if pos is not None and self.pos_injection == 2:
            x1d = self.pos_emb1d_2nd(x1d, pos)
        if self.layer == 3:
            x1d = self._conv_bn_relu(x1d, self.attention_second[0], self.attention_second[1])
            last = self.attention_third[0]
        else:
            last = self.attention_second[0]
        need_logits = attention_loss
        if need_logits:
            logits = ops.conv2d(x1d, self._weight4(last), last.bias, padding=(last.padding[0], 0))
            att = ops.affine_act(logits, None, None, None, 2)
        else:
            logits = None
            att = ops.conv2d(x1d, self._weight4(last), last.bias, padding=(last.padding[0], 0), act=2)   # sigmoid fused
        att = ops.interpolate(att, size=(H, 1), mode='bilinear')             # F.interpolate(mode='linear') along the rows
        out = ops.mul_rows(out, att)
        if return_attention:
            return out, att.squeeze(3)
        if attention_loss:
            return out, logits.squeeze(3)
        return out
