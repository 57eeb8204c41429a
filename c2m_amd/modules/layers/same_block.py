"""SameBlock2d / SameBlockTwoConv2d / SameBlock3d (reference: src/modules/layers/same_block.py:5-68)."""
from torch import nn

from .common import conv_module, batch_norm_module, instance_norm_module, pad_triple


class SameBlock2d(nn.Module):
    """conv -> InstanceNorm2d(affine) -> LeakyReLU(0.2)"""

    def __init__(self, in_features, out_features, kernel_size=3, stride=1, padding=1, padding_mode='zeros',
                 use_norm=True):
        super().__init__()
        self.conv = nn.Conv2d(in_channels=in_features, out_channels=out_features, stride=stride,
                              kernel_size=kernel_size, padding=padding, groups=1, padding_mode=padding_mode)
        self.norm = nn.InstanceNorm2d(out_features, affine=True)
        self.use_norm = use_norm

    def forward(self, x):
        if not self.use_norm:
            return conv_module(x, self.conv, act="lrelu")
        return instance_norm_module(conv_module(x, self.conv), self.norm, act="lrelu", private_input=True)   # (the conv output is a temporary)


class SameBlockTwoConv2d(nn.Module):
    """conv -> InstanceNorm2d(affine) -> LeakyReLU -> conv2"""

    def __init__(self, in_features, out_features, kernel_size=3, stride=1, padding=1, padding_mode='zeros',
                 use_norm=True):
        super().__init__()
        self.conv = nn.Conv2d(in_channels=in_features, out_channels=out_features, stride=stride,
                              kernel_size=kernel_size, padding=padding, groups=1, padding_mode=padding_mode)
        self.conv2 = nn.Conv2d(in_channels=out_features, out_channels=out_features, stride=stride,
                               kernel_size=kernel_size, padding=padding, groups=1, padding_mode=padding_mode)
        self.use_norm = use_norm
        if self.use_norm:
            self.norm = nn.InstanceNorm2d(out_features, affine=True)

    def forward(self, x):
        if self.use_norm:
            y = instance_norm_module(conv_module(x, self.conv), self.norm, act="lrelu", private_input=True)
        else:
            y = conv_module(x, self.conv, act="lrelu")
        return conv_module(y, self.conv2)


class SameBlock3d(nn.Module):
    """ReflectionPad3d -> Conv3d -> BatchNorm3d -> LeakyReLU"""

    def __init__(self, in_features, out_features, kernel_size=3, stride=1, padding=1, padding_mode='zeros',
                 use_norm=True):
        super().__init__()
        if padding_mode != "reflect":
            raise NotImplementedError(f"SameBlock3d padding_mode {padding_mode}")  # same_block.py:54-57
        self.pad_conv = nn.ReflectionPad3d(padding)
        self.conv = nn.Conv3d(in_channels=in_features, out_channels=out_features, stride=stride,
                              kernel_size=kernel_size, padding=0, groups=1, padding_mode=padding_mode)
        self.norm = nn.BatchNorm3d(out_features, affine=True)
        self.use_norm = use_norm
        self._pad3 = pad_triple(padding)

    def forward(self, x, dgrad_channels=None):
        """dgrad_channels: see ops.conv (x = cat([features, a tensor without grad]))."""
        if not self.use_norm:
            return conv_module(x, self.conv, act="lrelu", padding=self._pad3, padding_mode="reflect",
                               dgrad_channels=dgrad_channels)
        y = conv_module(x, self.conv, padding=self._pad3, padding_mode="reflect", dgrad_channels=dgrad_channels)
        return batch_norm_module(y, self.norm, act="lrelu", private_input=True)
