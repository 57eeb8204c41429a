"""OcclusionAwareGenerator (reference: src/modules/generator/generator.py:12-158).

SPADE path: the first frame is warped by the dense flow (HIP flow_warp), embedded together with flow/occlusion by the
FlowEmbedder, and modulates three ResidualSpadeBlocks.  Non-SPADE path: encoder features are warped with the
full-resolution flow bilinearly shrunk but NOT rescaled (deform_input mis-reads NCHW as NHWC, generator.py:81-86) --
kept as is."""
import torch
import torch.nn as nn

from ... import ops
from ..layers.residual_block import ResidualBlock, ResidualSpadeBlock
from ..layers.same_block import SameBlock2d
from ..layers.down_block import DownBlock2d
from ..layers.up_block import UpBlock2d
from ..layers.common import conv_module
from .flowembedder import FlowEmbedder


class OcclusionAwareGenerator(nn.Module):
    def __init__(self, model_params, flow_params, input_channel, dataset):
        super().__init__()
        self.input_channel = input_channel
        self.block_expansion = model_params["block_expansion"]
        self.num_down_blocks = model_params["num_down_blocks"]
        self.max_expansion = model_params["max_expansion"]
        self.num_bottleneck_blocks = model_params["num_bottleneck_blocks"]
        self.padding_mode = model_params["padding_mode"]
        self.use_spade = model_params["use_spade"]
        self.use_skip = model_params["use_skip"]
        self.spade_params = None
        self.flow_params = flow_params
        self.dataset = dataset
        nd, pm = self.num_down_blocks, self.padding_mode

        def width(level):
            return min(self.max_expansion, self.block_expansion * (2 ** level))

        def encoder():
            return [DownBlock2d(width(i), width(i + 1), kernel_size=4, stride=2, padding=1, padding_mode=pm)
                    for i in range(nd)]

        self.first = SameBlock2d(self.input_channel, self.block_expansion, kernel_size=7, padding=3, padding_mode=pm)
        self.down_blocks = nn.ModuleList(encoder())
        if "kitti" in self.dataset:
            self.first_warped = SameBlock2d(self.input_channel, self.block_expansion, kernel_size=7, padding=3,
                                            padding_mode=pm)
            self.down_blocks_warped = nn.Sequential(*encoder())
            self.pre_decode = nn.Sequential(SameBlock2d(width(nd) * 2, width(nd), kernel_size=3, padding=1,
                                                        padding_mode=pm))
        ups = []
        for i in range(nd):
            cin, cout = width(nd - i), width(nd - i - 1)
            if self.use_spade:
                ups.append(ResidualSpadeBlock(cond_dims=self.get_cond_dims(nd - i), in_planes=cin, out_planes=cout,
                                              kernel_size=3, padding=1, spade_params=self.spade_params))
            else:
                ups.append(UpBlock2d(cin, cout, kernel_size=3, padding=1, padding_mode=pm, reshape_3d=False,
                                     input_2d=True))
        self.up_blocks = nn.ModuleList(ups)
        self.middle = nn.Sequential(*[ResidualBlock(width(nd), width(nd), kernel_size=3, padding=1)
                                      for _ in range(self.num_bottleneck_blocks)])
        self.final = nn.Sequential(nn.Conv2d(self.block_expansion, 3, kernel_size=7, padding=3), nn.Sigmoid())
        if self.use_spade:
            self.upsample = nn.Upsample(scale_factor=2, mode="bilinear")
            self.flowembedder = FlowEmbedder(self.flow_params)

    @staticmethod
    def deform_input(inp, optical_flow):
        h, w = inp.shape[2:]
        if tuple(optical_flow.shape[2:]) != (h, w):
            # flow magnitudes are NOT rescaled here (reference behaviour)
            optical_flow = ops.resize_bilinear(optical_flow.detach(), (h, w)) if not optical_flow.requires_grad \
                else _resize_with_grad(optical_flow, (h, w))
        return ops.flow_warp(inp, optical_flow)

    def apply_optical(self, input_ref=None, optical_flow=None, occlusion_map=None):
        warped = self.deform_input(input_ref, optical_flow)
        if occlusion_map is None:
            return warped
        if warped.shape[2:] != occlusion_map.shape[2:]:
            occlusion_map = _resize_with_grad(occlusion_map, warped.shape[2:])
        return warped * occlusion_map

    def get_cond_dims(self, num_downs=0):
        num_downs = min(num_downs, self.flow_params["num_down_blocks"])
        return [min(self.max_expansion, self.block_expansion * (2 ** num_downs))]

    def get_cond_maps(self, label):
        return [[m] for m in self.flowembedder(label)]

    def forward(self, first_frame, flow, occlusion_map):
        nd = self.num_down_blocks
        if self.use_spade:
            img_warp = ops.flow_warp(first_frame, flow)
            cond = self.get_cond_maps(torch.cat([img_warp, flow, occlusion_map], dim=1))
        out = self.first(first_frame)
        for blk in self.down_blocks:
            out = blk(out)
        if not self.use_spade:
            out = self.apply_optical(input_ref=out, optical_flow=flow, occlusion_map=occlusion_map)
        for blk in self.middle:
            out = blk(out)
        if "kitti" in self.dataset:
            xw = self.first_warped(ops.flow_warp(first_frame, flow))
            for blk in self.down_blocks_warped:
                xw = blk(xw)
            occ = occlusion_map
            if xw.shape[2:] != occ.shape[2:]:
                occ = _resize_with_grad(occ, xw.shape[2:])
            out = self.pre_decode[0](torch.cat([out, xw * occ], dim=1))
        for i, blk in enumerate(self.up_blocks):
            if self.use_spade:
                c = cond[nd - i]
                if out.shape[-2:] != c[0].shape[-2:]:
                    raise NotImplementedError("generator feature/conditioning size mismatch")
                out = ops.upsample2x(blk(out, *c))
            else:
                out = blk(out)
        if out.shape[-2:] != first_frame.shape[-2:]:
            raise NotImplementedError("generator output size mismatch (input extents must be divisible by 8)")
        return conv_module(out, self.final[0], act="sigmoid")


def _resize_with_grad(x, size):
    """Differentiable bilinear resize for the rarely used non-SPADE / kitti branches (torch device op)."""
    return torch.nn.functional.interpolate(x, size=tuple(size), mode="bilinear")
