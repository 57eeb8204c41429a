"""FlowEmbedder: U-Net over [warped image, flow, occlusion] producing the SPADE conditioning maps
(reference: src/modules/generator/flowembedder.py:9-81)."""
import torch
from torch import nn

from ..layers.same_block import SameBlock2d
from ..layers.down_block import DownBlock2d
from ..layers.up_block import UpBlock2d


class FlowEmbedder(nn.Module):
    def __init__(self, model_params):
        super().__init__()
        self.input_channel = model_params["input_channel"]
        self.block_expansion = model_params["block_expansion"]
        self.num_down_blocks = model_params["num_down_blocks"]
        self.max_expansion = model_params["max_expansion"]
        self.padding_mode = model_params["padding_mode"]
        self.use_decoder = model_params["use_decoder"]
        nd, pm = self.num_down_blocks, self.padding_mode
        self.conv_first = SameBlock2d(self.input_channel, self.block_expansion, kernel_size=3, padding=1,
                                      padding_mode=pm, use_norm=False)
        ch = [min(self.max_expansion, self.block_expansion * (2 ** i)) for i in range(nd + 1)]
        self.down_blocks = nn.ModuleList([DownBlock2d(ch[i], ch[i + 1], kernel_size=4, stride=2, padding=1,
                                                      padding_mode=pm) for i in range(nd)])
        ups = []
        if self.use_decoder:
            for i in reversed(range(nd)):   # construction order of the reference (deepest first), stored shallow-first
                cin = ch[i + 1] * (2 if i != nd - 1 else 1)
                ups.append(UpBlock2d(cin, ch[i], kernel_size=3, stride=1, padding=1, padding_mode=pm, reshape_3d=False,
                                     input_2d=True))
        self.up_blocks = nn.ModuleList(ups[::-1])

    def forward(self, x):
        if x is None:
            return None
        nd = self.num_down_blocks
        enc = [self.conv_first(x)]
        for blk in self.down_blocks:
            enc.append(blk(enc[-1]))
        if not self.use_decoder:
            return enc
        dec, cur = [], enc[-1]
        for i in reversed(range(nd)):
            if i != nd - 1:
                if cur.shape[-2:] != enc[i + 1].shape[-2:]:
                    raise NotImplementedError("FlowEmbedder skip/upsample size mismatch (odd input extents)")
                cur = torch.cat([cur, enc[i + 1]], dim=1)
            cur = self.up_blocks[i](cur)
            dec.append(cur)
        # deepest encoder map followed by decoder maps, returned finest first (flowembedder.py:80-81)
        return ([enc[-1]] + dec)[::-1]
