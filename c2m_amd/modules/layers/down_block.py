"""DownBlock2d / DownBlock3d (reference: src/modules/layers/down_block.py:5-48): conv -> BatchNorm -> LeakyReLU(0.2).
Same constructor signatures and state_dict keys (conv.*, norm.*); the conv (with its reflection padding folded into the
gather) and the fused BN+LeakyReLU run as HIP kernels."""
from torch import nn

from .common import conv_module, batch_norm_module, pad_triple


class DownBlock2d(nn.Module):
    def __init__(self, in_features, out_features, kernel_size=(3, 3), stride=(1, 1), padding=1, padding_mode='zeros',
                 use_norm=True):
        super().__init__()
        self.conv = nn.Conv2d(in_channels=in_features, out_channels=out_features, stride=stride,
                              kernel_size=kernel_size, padding=padding, groups=1, padding_mode=padding_mode)
        self.norm = nn.BatchNorm2d(out_features, affine=True)
        self.use_norm = use_norm

    def forward(self, x):
        if not self.use_norm:
            return conv_module(x, self.conv, act="lrelu")
        return batch_norm_module(conv_module(x, self.conv), self.norm, act="lrelu", private_input=True)   # (the conv output is a temporary)


class DownBlock3d(nn.Module):
    def __init__(self, in_features, out_features, kernel_size=[3, 3, 3], stride=[1, 1, 1], padding=[1, 1, 1],
                 padding_mode='zeros', use_norm=True):
        super().__init__()
        if padding_mode not in ("reflect",):
            # the reference only defines pad_conv for reflect/replicate (down_block.py:34-37); replicate is unused
            raise NotImplementedError(f"DownBlock3d padding_mode {padding_mode}")
        self.pad_conv = nn.ReflectionPad3d(padding)   # kept for module-tree parity; folded into the conv gather
        self.conv = nn.Conv3d(in_channels=in_features, out_channels=out_features, stride=stride,
                              kernel_size=kernel_size, padding=0, groups=1, padding_mode=padding_mode)
        self.norm = nn.BatchNorm3d(out_features, affine=True)
        self.use_norm = use_norm
        self._pad3 = pad_triple(padding)

    def forward(self, x):
        if not self.use_norm:
            return conv_module(x, self.conv, act="lrelu", padding=self._pad3, padding_mode="reflect")
        y = conv_module(x, self.conv, padding=self._pad3, padding_mode="reflect")
        return batch_norm_module(y, self.norm, act="lrelu", private_input=True)
