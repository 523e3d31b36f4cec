"""ASPP head of the reference (models/aspp.py), MI355X-native: same module tree / state_dict keys."""
import torch
import torch.nn as nn

from . import ops

_layer_ids = [100]


class _ASPPModule(nn.Module):
    """models/aspp.py:7-32: atrous conv -> BN -> ReLU as one fused node."""

    def __init__(self, inplanes, planes, kernel_size, padding, dilation, BatchNorm=nn.BatchNorm2d):
        super().__init__()
        self.atrous_conv = nn.Conv2d(inplanes, planes, kernel_size=kernel_size, stride=1, padding=padding, dilation=dilation, bias=False)
        self.bn = BatchNorm(planes)
        self.relu = nn.ReLU()
        nn.init.kaiming_normal_(self.atrous_conv.weight)
        self.bn.weight.data.fill_(1)
        self.bn.bias.data.zero_()

    def forward(self, x, groups=1):
        c = self.atrous_conv
        return ops.conv_bn_act(x, c.weight, self.bn, kind='conv', stride=1, dilation=c.dilation[0], padding=c.padding[0],
                               act=1, groups=groups)


class ASPP(nn.Module):
    """models/aspp.py:34-108."""
    _INPLANES = {'drn': 512, 'mobilenet': 320, 'densenet_a1': 128, 'densenet_a3': 512, 'mobilenet_a1': 24,
                 'mobilenet_a3': 112, 'resnet50_a1': 256, 'resnet50_a3': 1024, 'resnet50_a4': 2048}
    _DIL = {32: [1, 2, 6, 12], 16: [1, 6, 12, 18], 8: [1, 12, 24, 36]}

    def __init__(self, backbone, output_stride, BatchNorm=nn.BatchNorm2d):
        super().__init__()
        cin = self._INPLANES.get(backbone, 2048)
        if output_stride not in self._DIL:
            raise NotImplementedError
        d = self._DIL[output_stride]
        self.aspp1 = _ASPPModule(cin, 256, 1, 0, d[0], BatchNorm)
        self.aspp2 = _ASPPModule(cin, 256, 3, d[1], d[1], BatchNorm)
        self.aspp3 = _ASPPModule(cin, 256, 3, d[2], d[2], BatchNorm)
        self.aspp4 = _ASPPModule(cin, 256, 3, d[3], d[3], BatchNorm)
        self.global_avg_pool = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), nn.Conv2d(cin, 256, 1, stride=1, bias=False),
                                             BatchNorm(256), nn.ReLU())
        self.conv1 = nn.Conv2d(1280, 256, 1, bias=False)
        self.bn1 = BatchNorm(256)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(0.5)
        self._drop_id = _layer_ids[0]
        _layer_ids[0] += 1
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                torch.nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def forward(self, x, groups=1):
        gp = self.global_avg_pool
        x5 = ops.conv_bn_act(ops.global_avg_pool(x), gp[1].weight, gp[2], act=1, groups=groups)
        x5 = ops.interpolate(x5, size=x.shape[2:], mode='bilinear', align_corners=True)
        y = ops.concat([self.aspp1(x, groups), self.aspp2(x, groups), self.aspp3(x, groups), self.aspp4(x, groups), x5])
        y = ops.conv_bn_act(y, self.conv1.weight, self.bn1, act=1, groups=groups)
        return ops.dropout(y, self.dropout.p, self.training, self._drop_id)


def build_aspp(backbone, output_stride, BatchNorm=nn.BatchNorm2d):
    return ASPP(backbone, output_stride, BatchNorm)
