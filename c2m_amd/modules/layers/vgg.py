"""Vgg19 feature slicer (reference: src/modules/layers/vgg.py:7-137).

torchvision is not a dependency here: the VGG-19 "E" feature stack is rebuilt with the same `features` indices so the
state_dict keys (relu1_1.0.weight ... relu5_4.34.bias, mean, std) match.  The reference downloads ImageNet weights
(`pretrained=True`, vgg.py:13); offline we initialise like torchvision does and `load_state_dict` accepts the real ones.
conv3x3+bias+ReLU is one implicit-GEMM launch; max-pools are a HIP kernel.  With `taps_only=True` (default when the
style loss is off) evaluation stops at relu5_1 -- the three convs after it are never consumed."""
import numpy as np
import torch
from torch import nn

from ... import ops

_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512]
_NAMES = ["relu1_1", "relu1_2", "relu2_1", "relu2_2", "relu3_1", "relu3_2", "relu3_3", "relu3_4",
          "relu4_1", "relu4_2", "relu4_3", "relu4_4", "relu5_1", "relu5_2", "relu5_3", "relu5_4"]


class Vgg19(torch.nn.Module):
    def __init__(self, requires_grad=False, stop_after="relu5_4"):
        super().__init__()
        idx, cin, ni = 0, 3, 0
        pending_pool = None
        self._plan = []   # (slice name, [(kind, features index)])
        for v in _CFG:
            if v == "M":
                pending_pool = idx
                idx += 1
                continue
            seq = nn.Sequential()
            steps = []
            if pending_pool is not None:
                seq.add_module(str(pending_pool), nn.MaxPool2d(2, 2))
                steps.append(("pool", pending_pool))
                pending_pool = None
            conv = nn.Conv2d(cin, v, 3, padding=1)
            nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")
            nn.init.constant_(conv.bias, 0)
            seq.add_module(str(idx), conv)
            seq.add_module(str(idx + 1), nn.ReLU(inplace=True))
            steps.append(("conv", idx))
            setattr(self, _NAMES[ni], seq)
            self._plan.append((_NAMES[ni], steps))
            idx += 2
            cin = v
            ni += 1
        self.mean = torch.nn.Parameter(data=torch.Tensor(np.array([0.485, 0.456, 0.406]).reshape((1, 3, 1, 1))),
                                       requires_grad=False)
        self.std = torch.nn.Parameter(data=torch.Tensor(np.array([0.229, 0.224, 0.225]).reshape((1, 3, 1, 1))),
                                      requires_grad=False)
        self.stop_after = stop_after
        if not requires_grad:
            for param in self.parameters():
                param.requires_grad = False

    def forward(self, x, tap_targets=None, need=()):
        """tap_targets (optional): {slice name: target feature map without gradient}; those slices also return mean|y - target|
        (the perceptual loss's feature L1, losses/losses.py:60-65) under out["l1"][name], with the L1 gradient, the sum with the
        next conv's data gradient and the ReLU mask fused into one backward pass (ops.conv_relu_tap)."""
        x = (x - self.mean) / self.std
        out = {}
        l1 = {}
        flat = []                              # [(kind, features index, slice the op closes or None)]
        for name, steps in self._plan:
            for k, (kind, i) in enumerate(steps):
                flat.append((kind, i, name, k == len(steps) - 1))
            if name == self.stop_after:
                break
        j = 0
        while j < len(flat):
            kind, i, name, closes = flat[j]
            if kind == "pool":
                x = ops.maxpool2x2(x)
            else:
                c = getattr(self, name)._modules[str(i)]
                fuse_pool = tap_targets is not None and name not in tap_targets and name not in need and j + 1 < len(flat) and \
                    flat[j + 1][0] == "pool"
                if tap_targets is not None and name in tap_targets:
                    x, l1[name] = ops.conv_relu_tap(x, c.weight, c.bias, tap_targets[name])
                elif fuse_pool:
                    # conv -> ReLU -> MaxPool2d with nobody else reading the ReLU output (training path: only the taps are
                    # consumed): one node, ReLU backward inside the pool backward.  out[name] is not produced then.
                    x = ops.conv_relu_pool(x, c.weight, c.bias)
                    j += 1
                    closes = False
                else:
                    x = ops.conv(x, c.weight, c.bias, stride=1, padding=1, padding_mode="zeros", act="relu")
            if closes:
                out[name] = x
            j += 1
        if tap_targets is not None:
            out["l1"] = l1
        return out
