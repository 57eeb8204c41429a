"""SparseMotionFeatureEncoder (reference: src/modules/motion_estimator/sparse_encoder.py:6-28)."""
from torch import nn

from ..layers.down_block import DownBlock3d


class SparseMotionFeatureEncoder(nn.Module):
    def __init__(self, params):
        super().__init__()
        self.input_channel = params["in_channel"]
        self.block_expansion = params["block_expansion"]
        self.num_down_blocks = params["num_down_blocks"]
        self.max_expansion = params["max_expansion"]
        self.padding_mode = params["padding_mode"]
        widths = [self.input_channel] + [min(self.max_expansion, self.block_expansion * (2 ** i))
                                         for i in range(self.num_down_blocks)]
        self.down_blocks = nn.ModuleList([
            DownBlock3d(in_features=widths[i], out_features=widths[i + 1], kernel_size=[3, 4, 4], stride=[1, 2, 2],
                        padding=1, padding_mode=self.padding_mode) for i in range(self.num_down_blocks)])

    def forward(self, sparse_motion):
        out, x = {}, sparse_motion
        for i, blk in enumerate(self.down_blocks):
            x = blk(x)
            out[f"enco_sparse_{i}"] = x
        return out
