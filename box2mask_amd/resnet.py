"""BasicBlock / ResNetBase with the attribute names of /root/reference/models/resnet.py:46-181
(only what Box2Mask's SelectionNet uses; the ResNet14..101 / ResFieldNet variants are unused there).

The block's forward is the reference's conv-BN-ReLU-conv-BN (+1x1conv-BN residual) add ReLU
(resnet.py:70-83) with BN+ReLU and BN+add+ReLU fused into single elementwise launches.
"""
from __future__ import annotations

from torch import nn

from . import nn as ME


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, dimension=-1,
                 expand_coordinates=False):
        super().__init__()
        assert dimension > 0
        self.conv1 = ME.MinkowskiConvolution(inplanes, planes, kernel_size=3, stride=stride, dilation=dilation,
                                             dimension=dimension, expand_coordinates=expand_coordinates)
        self.norm1 = ME.MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = ME.MinkowskiConvolution(planes, planes, kernel_size=3, stride=1, dilation=dilation,
                                             dimension=dimension)
        self.norm2 = ME.MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.relu = ME.MinkowskiReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        out = self.conv1(x)
        ck = ME.count_key_of(out)
        out = out.new(self.norm1.apply_bn(out.F, relu=True, count_key=ck, defer_counter=True))
        out = self.conv2(out)
        if self.downsample is not None:
            res = self.downsample[0](x)
            residual = self.downsample[1].apply_bn(res.F, count_key=ck, defer_counter=True)
        else:
            residual = x.F
        return out.new(self.norm2.apply_bn(out.F, residual=residual, relu=True, count_key=ck, defer_counter=True))


class ResNetBase(nn.Module):
    BLOCK = None
    LAYERS = ()
    INIT_DIM = 64
    PLANES = (64, 128, 256, 512)

    def __init__(self, in_channels, out_channels, D=3, expand_coordinates=False):
        nn.Module.__init__(self)
        self.D = D
        self.expand_coordinates = expand_coordinates
        assert self.BLOCK is not None
        self.network_initialization(in_channels, out_channels, D)
        self.weight_initialization()

    def weight_initialization(self):
        # resnet.py:139-146: Kaiming fan-out on MinkowskiConvolution kernels only (transposed
        # convolutions keep their default init); BN gamma=1, beta=0
        for m in self.modules():
            if isinstance(m, ME.MinkowskiConvolution):
                ME.kaiming_normal_(m.kernel, mode='fan_out', nonlinearity='relu')
            if isinstance(m, ME.MinkowskiBatchNorm):
                nn.init.constant_(m.bn.weight, 1)
                nn.init.constant_(m.bn.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1, bn_momentum=0.1, expand_coordinates=False):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                ME.MinkowskiConvolution(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride,
                                        dimension=self.D),
                ME.MinkowskiBatchNorm(planes * block.expansion),
            )
        layers = [block(self.inplanes, planes, stride=stride, dilation=dilation, downsample=downsample,
                        dimension=self.D, expand_coordinates=expand_coordinates)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, stride=1, dilation=dilation, dimension=self.D,
                                expand_coordinates=expand_coordinates))
        return nn.Sequential(*layers)
