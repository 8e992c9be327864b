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
        if self.norm1.fusable() and self.norm2.fusable() and (self.downsample is None or self.downsample[1].fusable()):
            # inference: every BatchNorm is a per-channel affine map that its convolution applies on the way out -- three
            # (four with a shortcut) launches per block instead of six (nine), no tensor written un-normalised
            out = self.conv1(x, fuse=(self.norm1, None, True))
            residual = x.F if self.downsample is None else self.downsample[0](x, fuse=(self.downsample[1], None, False)).F
            return self.conv2(out, fuse=(self.norm2, residual, True))
        # (x from here on: the alias conv1 hands back -- the residual / shortcut gradient is then summed with conv1's data
        # gradient inside the convolution kernel, not by an add kernel per block)
        out, x = self.conv1(x, passthrough=True)
        ck = ME.count_key_of(out)
        out = out.new(self.norm1.apply_bn(out.F, relu=True, count_key=ck, defer_counter=True))
        out = self.conv2(out)
        if self.downsample is not None:
            res = self.downsample[0](x)
            # norm2 and the shortcut's BatchNorm are independent and meet in the add: one paired operator (one apply, one
            # backward reduction, one SyncBN exchange per direction for both layers) where it applies
            y = ME.batch_norm_add_relu(self.norm2, out.F, self.downsample[1], res.F, relu=True, defer_counter=True,
                                       count_key=ck)
            if y is not None:
                return out.new(y)
            residual = self.downsample[1].apply_bn(res.F, count_key=ck, defer_counter=True)
        else:
            residual = x.F
        return out.new(self.norm2.apply_bn(out.F, residual=residual, relu=True, count_key=ck, defer_counter=True))


class ResNetBase(nn.Module):
    """Skeleton the U-Net derives from (resnet.py:86-181): subclasses set BLOCK / LAYERS / PLANES and build their
    layers in `network_initialization`; `_make_layer` stacks residual blocks and tracks `self.inplanes`."""
    BLOCK = None
    LAYERS = ()
    INIT_DIM = 64
    PLANES = (64, 128, 256, 512)

    def __init__(self, in_channels, out_channels, D=3, expand_coordinates=False):
        super().__init__()
        if self.BLOCK is None:
            raise TypeError('%s must define BLOCK' % type(self).__name__)
        self.D, self.expand_coordinates = D, expand_coordinates
        self.network_initialization(in_channels, out_channels, D)
        self.weight_initialization()

    def weight_initialization(self):
        """He-normal (fan-out) on every non-transposed convolution kernel, identity affine on every BatchNorm;
        transposed convolutions keep their constructor's uniform init (resnet.py:139-146 touches neither them nor
        biases)."""
        for module in self.modules():
            if type(module) is ME.MinkowskiConvolution:
                ME.kaiming_normal_(module.kernel, mode='fan_out', nonlinearity='relu')
            elif isinstance(module, ME.MinkowskiBatchNorm):
                nn.init.ones_(module.bn.weight)
                nn.init.zeros_(module.bn.bias)

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1, bn_momentum=0.1, expand_coordinates=False):
        """`blocks` residual blocks of width planes * block.expansion.  Only the first one may change the width or
        the stride; when it does, its shortcut is a 1x1 convolution + BatchNorm (state-dict names `downsample.0/.1`)."""
        width = planes * block.expansion
        shortcut = None
        if stride != 1 or self.inplanes != width:
            shortcut = nn.Sequential(
                ME.MinkowskiConvolution(self.inplanes, width, kernel_size=1, stride=stride, dimension=self.D),
                ME.MinkowskiBatchNorm(width))
        stack = []
        for index in range(blocks):
            first = index == 0
            stack.append(block(self.inplanes if first else width, planes, stride=stride if first else 1,
                               dilation=dilation, downsample=shortcut if first else None, dimension=self.D,
                               expand_coordinates=expand_coordinates))
        self.inplanes = width
        return nn.Sequential(*stack)
