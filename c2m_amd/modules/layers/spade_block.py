"""SpatiallyAdaptiveNorm (reference: src/modules/layers/spade_block.py:7-77).  The instance-norm statistics, the
x_hat*(1+gamma)+beta modulation and (optionally) the following LeakyReLU are one fused HIP kernel."""
import torch.nn.functional as F
from torch import nn

from ... import ops
from .common import conv_module
from .same_block import SameBlock2d


class SpatiallyAdaptiveNorm(nn.Module):
    def __init__(self, num_features, cond_dims, num_filters=128, kernel_size=3, bias_only=False,
                 interpolation='nearest'):
        super().__init__()
        padding = kernel_size // 2
        self.mlps = nn.ModuleList()
        self.bias_only = bias_only
        self.interpolation = interpolation
        if type(cond_dims) != list:
            cond_dims = [cond_dims]
        if not isinstance(num_filters, list):
            num_filters = [num_filters] * len(cond_dims)
        else:
            assert len(num_filters) >= len(cond_dims)
        for i, cond_dim in enumerate(cond_dims):
            mlp = []
            if num_filters[i] > 0:
                mlp += [SameBlock2d(cond_dim, num_filters[i], kernel_size, padding=padding, padding_mode="reflect",
                                    use_norm=False)]
            mlp_ch = cond_dim if num_filters[i] == 0 else num_filters[i]
            mlp += [nn.Conv2d(mlp_ch, num_features * 2, kernel_size, stride=1, padding=padding, padding_mode="reflect")]
            self.mlps.append(nn.Sequential(*mlp))
        self.norm = nn.InstanceNorm2d(num_features, affine=False)
        self.conditional = True

    def _gamma_beta(self, i, cond, size):
        if tuple(cond.shape[2:]) != tuple(size):
            cond = F.interpolate(cond, size=size, mode=self.interpolation)
        h = cond
        for layer in self.mlps[i]:
            h = layer(h) if isinstance(layer, SameBlock2d) else conv_module(h, layer)
        return h

    def forward(self, x, *cond_inputs, act=None, feeds=None, private_input=False, **_kwargs):
        live = [(i, c) for i, c in enumerate(cond_inputs) if c is not None]
        if len(live) != 1 or self.bias_only:
            raise NotImplementedError("SPADE with other than one conditional map / bias_only is not on the C2M path")
        i, cond = live[0]
        return ops.spade_norm_act(x, self._gamma_beta(i, cond, x.shape[2:]), act, self.norm.eps, feeds, private_input)
