"""conv -> BatchNorm -> ReLU blocks (support for rows R, M and ImageConv).

Same constructor signatures, sub-module names (``conv``, ``bn``) and initialisation as the reference
blocks (reference nn/conv.py:7-216) so state dicts are interchangeable; implemented once, generically,
instead of five times.  A block's own ``forward`` is the stock ATen composition (it carries the autograd graph of the
training step); the inference paths of the containers -- ``ImageConv.forward`` / ``forward_views``,
``VolumeConv.forward`` / ``forward_fused``, ``SharedMLP.forward`` -- run the blocks' convolutions and BatchNorms on
the HIP kernels of csrc/ (conv2d_wide.hip, conv3d*.hip, deconv3d.hip, edgeconv.hip, norm.hip) instead.
"""
from torch import nn
import torch.nn.functional as F

from .init import init_bn, init_uniform


class _ConvBlock(nn.Module):
    conv_type = None
    bn_type = None
    transposed = False

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, relu=True, bn=True,
                 bn_momentum=0.1, **kwargs):
        super(_ConvBlock, self).__init__()
        if self.transposed or self.conv_type is nn.Conv3d:
            assert stride in [1, 2]
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.stride = stride
        self.conv = self.conv_type(in_channels, out_channels, kernel_size, stride=stride, bias=(not bn), **kwargs)
        self.bn = self.bn_type(out_channels, momentum=bn_momentum) if bn else None
        self.relu = relu
        self.init_weights()

    def init_weights(self):
        init_uniform(self.conv)
        if self.bn is not None:
            init_bn(self.bn)

    def _crop(self, y, x):
        return y

    def forward(self, x):
        y = self._crop(self.conv(x), x)
        if self.bn is not None:
            y = self.bn(y)
        if self.relu:
            y = F.relu(y, inplace=True)
        return y


class Conv1d(_ConvBlock):
    conv_type, bn_type = nn.Conv1d, nn.BatchNorm1d

    def __init__(self, in_channels, out_channels, kernel_size, relu=True, bn=True, bn_momentum=0.1, **kwargs):
        # the reference Conv1d has no positional stride (nn/conv.py:18-19)
        stride = kwargs.pop("stride", 1)
        super(Conv1d, self).__init__(in_channels, out_channels, kernel_size, stride, relu, bn, bn_momentum,
                                     **kwargs)


class Conv2d(_ConvBlock):
    conv_type, bn_type = nn.Conv2d, nn.BatchNorm2d


class Conv3d(_ConvBlock):
    conv_type, bn_type = nn.Conv3d, nn.BatchNorm3d


class Deconv2d(_ConvBlock):
    conv_type, bn_type, transposed = nn.ConvTranspose2d, nn.BatchNorm2d, True

    def _crop(self, y, x):
        # stride-2 transposed conv output is cropped to exactly twice the input (reference nn/conv.py:160-162)
        if self.stride == 2:
            h, w = x.shape[2:]
            y = y[:, :, :2 * h, :2 * w].contiguous()
        return y


class Deconv3d(_ConvBlock):
    conv_type, bn_type, transposed = nn.ConvTranspose3d, nn.BatchNorm3d, True
