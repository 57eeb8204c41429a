"""ResidualBlock / ResidualSpadeBlock (reference: src/modules/layers/residual_block.py:6-71).  The explicit
ReflectionPad2d before each conv is folded into the conv gather; norm+activation are fused kernels."""
import torch.nn as nn

from .common import conv_module, batch_norm_module, feeds_conv
from .spade_block import SpatiallyAdaptiveNorm


class ResidualBlock(nn.Module):
    def __init__(self, in_planes, out_planes, kernel_size, padding):
        super().__init__()
        self.padding = nn.ReflectionPad2d(padding)
        self.conv1 = nn.Conv2d(in_channels=in_planes, out_channels=out_planes, kernel_size=kernel_size, padding=0)
        self.conv2 = nn.Conv2d(in_channels=out_planes, out_channels=out_planes, kernel_size=kernel_size, padding=0)
        self.norm1 = nn.BatchNorm2d(in_planes, affine=True)
        self.norm2 = nn.BatchNorm2d(out_planes, affine=True)
        self._pad = padding

    def forward(self, x):
        # (each norm result is read by the convolution behind it and by nothing else: feeds -> NC8-only output where that pays)
        out = batch_norm_module(x, self.norm1, act="relu", feeds=feeds_conv(self.conv1, self._pad, "reflect"))
        out = conv_module(out, self.conv1, padding=self._pad, padding_mode="reflect")
        out = batch_norm_module(out, self.norm2, act="relu", feeds=feeds_conv(self.conv2, self._pad, "reflect"), private_input=True)
        out = conv_module(out, self.conv2, padding=self._pad, padding_mode="reflect")
        return out + x


class ResidualSpadeBlock(nn.Module):
    def __init__(self, cond_dims, in_planes, out_planes, kernel_size, padding, spade_params):
        super().__init__()
        self.cond_dims = cond_dims
        self.spade_params = spade_params
        self.padding = nn.ReflectionPad2d(padding)
        self.conv1 = nn.Conv2d(in_channels=in_planes, out_channels=out_planes, kernel_size=kernel_size, padding=0)
        self.conv2 = nn.Conv2d(in_channels=out_planes, out_channels=out_planes, kernel_size=kernel_size, padding=0)
        self.norm1 = SpatiallyAdaptiveNorm(in_planes, cond_dims)
        self.norm2 = SpatiallyAdaptiveNorm(out_planes, cond_dims)
        self.learned_shortcut = (in_planes != out_planes)
        if self.learned_shortcut:
            self.conv_s = nn.Conv2d(in_planes, out_planes, kernel_size=1, bias=False)
            self.norm_s = SpatiallyAdaptiveNorm(in_planes, cond_dims)
        self._pad = padding

    def forward(self, x, *cond_inputs):
        dx = self.norm1(x, *cond_inputs, act="lrelu", feeds=feeds_conv(self.conv1, self._pad, "reflect"))
        dx = conv_module(dx, self.conv1, padding=self._pad, padding_mode="reflect")
        dx = self.norm2(dx, *cond_inputs, act="lrelu", feeds=feeds_conv(self.conv2, self._pad, "reflect"), private_input=True)
        dx = conv_module(dx, self.conv2, padding=self._pad, padding_mode="reflect")
        if self.learned_shortcut:
            x_s = self.norm_s(x, *cond_inputs, act="lrelu", feeds=feeds_conv(self.conv_s))
            return dx + conv_module(x_s, self.conv_s)
        return dx
