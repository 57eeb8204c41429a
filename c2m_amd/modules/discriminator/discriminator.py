"""Patch discriminators + LSGAN criterion (reference: src/modules/discriminator/discriminator.py:10-135)."""
import torch
import torch.nn as nn

from ... import ops
from ..layers.down_block import DownBlock2d


def weights_init(m):
    name = m.__class__.__name__
    if 'Conv' in name and hasattr(m, 'weight'):
        m.weight.data.normal_(0.0, 0.02)
    elif 'BatchNorm2d' in name:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)


def define_d(input_nc, ndf, n_layers_d, num_d=1, padding_mode="zeros"):
    net = MultiScaleDiscriminator(input_nc, ndf, n_layers_d, num_d, padding_mode)
    net.apply(weights_init)
    return net


class MultiScaleDiscriminator(nn.Module):
    def __init__(self, input_nc, ndf, n_layers_d, num_d, padding_mode):
        super().__init__()
        self.scales = num_d
        self.discs = nn.ModuleDict({str(s).replace('.', '-'): Discriminator(num_channels=input_nc, block_expansion=ndf,
                                                                            num_blocks=n_layers_d, padding_mode=padding_mode)
                                    for s in range(self.scales)})

    def forward(self, x):
        out = {}
        for scale, disc in self.discs.items():
            scale = str(scale).replace('-', '.')
            out['feature_maps_' + scale], out['prediction_map_' + scale] = disc(x)
        return out


class Discriminator(nn.Module):
    """4x DownBlock2d (k4 s2 reflect, BN) + spectral-normalised 1x1 conv."""

    def __init__(self, num_channels=3, block_expansion=64, num_blocks=4, max_features=512, sn=True,
                 padding_mode="zeros"):
        super().__init__()
        self.down_blocks = nn.ModuleList([
            DownBlock2d(num_channels if i == 0 else min(max_features, block_expansion * (2 ** i)),
                        min(max_features, block_expansion * (2 ** (i + 1))), kernel_size=4, stride=2, padding=1,
                        padding_mode=padding_mode, use_norm=True) for i in range(num_blocks)])
        self.conv = nn.Conv2d(self.down_blocks[-1].conv.out_channels, out_channels=1, kernel_size=1)
        if sn:
            self.conv = nn.utils.spectral_norm(self.conv)
        self._sn = sn

    def forward(self, x):
        feats = []
        for blk in self.down_blocks:
            x = blk(x)
            feats.append(x)
        if self._sn:
            # run the spectral-norm pre-forward hook (power iteration on weight_u/v, sets .weight = weight_orig / sigma)
            for hook in self.conv._forward_pre_hooks.values():
                hook(self.conv, (x,))
        return feats, ops.conv(x, self.conv.weight, self.conv.bias)


class GANLoss(nn.Module):
    """LSGAN (MSE to a constant label) evaluated on the LAST batch element only, as in the reference (:134-135)."""

    def __init__(self, use_lsgan=True, target_real_label=1.0, target_fake_label=0.0, tensor=torch.FloatTensor):
        super().__init__()
        self.real_label, self.fake_label = target_real_label, target_fake_label
        self.real_label_var = self.fake_label_var = None
        self.Tensor = tensor
        self.loss = nn.MSELoss() if use_lsgan else nn.BCELoss()

    def get_target_tensor(self, input_tensor, target_is_real):
        return torch.full_like(input_tensor, self.real_label if target_is_real else self.fake_label)

    def __call__(self, input_tensor, target_is_real):
        if isinstance(input_tensor[0], list):
            return sum(self.loss(i[-1], self.get_target_tensor(i[-1], target_is_real)) for i in input_tensor)
        pred = input_tensor[-1]
        return self.loss(pred, self.get_target_tensor(pred, target_is_real))
