"""UpBlock2d (reference: src/modules/layers/up_block.py:5-27): bilinear x2 -> conv -> BatchNorm2d -> LeakyReLU on
time-folded frames.  `main` keeps the reference's Sequential layout so the keys stay main.1.* / main.2.*."""
from torch import nn

from ... import ops
from .common import conv_module, batch_norm_module, feeds_conv, fold_time, unfold_time


class UpBlock2d(nn.Module):
    def __init__(self, in_features, out_features, kernel_size=3, stride=1, padding=1, padding_mode='zeros',
                 reshape_3d=True, input_2d=False):
        super().__init__()
        self.main = nn.Sequential(
            nn.Upsample(scale_factor=2, mode="bilinear"),
            nn.Conv2d(in_features, out_features, kernel_size, stride, padding, padding_mode=padding_mode),
            nn.BatchNorm2d(out_features),
            nn.LeakyReLU(0.2, inplace=True))
        self.reshape_3d = reshape_3d
        self.input_2d = input_2d

    def forward(self, x):
        flat = x if self.input_2d else fold_time(x)
        y = conv_module(ops.upsample2x(flat, feeds=feeds_conv(self.main[1])), self.main[1])     # (the up-sampled map: read by this conv only)
        y = batch_norm_module(y, self.main[2], act="lrelu", private_input=True)       # (y: this block's conv output, read here only)
        # the reference hard-codes 5 predicted frames here (up_block.py:25)
        return unfold_time(y, 5) if self.reshape_3d else y
