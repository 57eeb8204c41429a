"""AppearanceEncoder (reference: src/modules/appearance_encoder/appearance_encoder.py:8-78).

Six stride-2 4x4 reflect DownBlock2d on the first frame(s) + per-object RoI-align features.  The reference's RoI
quirks (box column order, t_in=2 batch-index interleave, :57-62 and :72-76) are reproduced verbatim."""
import torch
from torch import nn

from ..layers.same_block import SameBlock2d
from ..layers.down_block import DownBlock2d
from ... import ops


class AppearanceEncoder(nn.Module):
    def __init__(self, train_params, block_expansion, num_down_blocks, max_expansion, pooling_after, padding_mode,
                 pool_size, scale_factor, image_channel, seg_channel_bg, seg_channel_fg, instance_channel,
                 flow_channel, occlusion_channel):
        super().__init__()
        self.train_params = train_params
        self.pool_size = pool_size
        t_in = train_params["num_input_frames"]
        self.h_appearance_map = int(train_params["input_size"][0] / (2 ** num_down_blocks) * scale_factor)
        self.w_appearance_map = int(train_params["input_size"][1] / (2 ** num_down_blocks) * scale_factor)
        blocks = []
        for i in range(num_down_blocks):
            width_in = min(max_expansion, block_expansion * (2 ** (i - 1))) * t_in
            width_out = min(max_expansion, block_expansion * (2 ** i))
            if i == 0:
                width_in = (image_channel + seg_channel_bg + seg_channel_fg + instance_channel) * t_in + \
                           (flow_channel + occlusion_channel) * (t_in - 1)
                width_out = block_expansion * t_in
            elif i != num_down_blocks - 1:
                width_out = width_out * t_in
            blocks.append(DownBlock2d(in_features=width_in, out_features=width_out, kernel_size=4, stride=2, padding=1,
                                      padding_mode=padding_mode, use_norm=True))
        self.h_flatten_appearance = self.h_appearance_map * self.w_appearance_map * width_out
        roi_in = block_expansion * (2 ** (pooling_after - 1))
        roi_out = block_expansion * (2 ** pooling_after)
        roi_blocks = [SameBlock2d(in_features=roi_in, out_features=roi_out * 2, kernel_size=self.pool_size, stride=1,
                                  padding=0, padding_mode=padding_mode, use_norm=False),
                      nn.Flatten(),
                      nn.Linear(in_features=roi_out * 2, out_features=roi_out * 2)]
        self.roi_align_regressor = nn.Linear(in_features=roi_out * 2, out_features=roi_out)
        self.fuse_appearance_roi = nn.Linear(in_features=roi_out + self.h_flatten_appearance, out_features=roi_out)
        self.down_blocks = nn.ModuleList(blocks)
        self.roi_align_blocks = nn.Sequential(*roi_blocks)
        self.spatial_scale = (1 / scale_factor) * 2 ** pooling_after
        self.pooling_after = pooling_after
        # True (set by GeneratorFullModel): `objects_feature` is handed on WITHOUT joining the auxiliary stream -- its only reader, the
        # object GNN, continues there, and the model joins in front of the losses (ops.aux_branch)
        self.defer_aux_join = False

    def forward(self, input_dict):
        gnn = input_dict["tracking_gnn"]
        t_in = self.train_params["num_input_frames"]
        out = {}
        x = input_dict["first_frame"]
        last = len(self.down_blocks) - 1
        for i, blk in enumerate(self.down_blocks):
            x = blk(x)
            out["app_encoded" if i == last else f"enco{i}"] = x
        # the RoI head -- a dozen small launches whose only consumer is the object GNN -- goes to the auxiliary stream in training
        # (ops.aux_branch: it runs next to the motion encoders' convolutions; the GNN continues on that stream, the model joins it in
        # front of the losses or the raster)
        feat, enc = out[f"enco{self.pooling_after - 1}"], out["app_encoded"]
        with ops.aux_branch(*((feat, enc) if self.training and torch.is_grad_enabled() else ()), part="roi"):
            boxes = torch.cat([gnn.batch.unsqueeze(1).repeat_interleave(t_in, dim=0),
                               torch.cat(torch.unbind(gnn.source_frames_nodes_roi_padded, dim=1), dim=0)], dim=1)
            pooled_src = torch.cat(feat.chunk(t_in, 1), dim=0)
            obj = ops.roi_align(pooled_src, boxes, self.pool_size, spatial_scale=1 / self.spatial_scale)
            obj = self.roi_align_regressor(self.roi_align_blocks(obj))
            # == torch.repeat_interleave(app_encoded.flatten(1), gnn.num_real_nodes * t_in, dim=0) of the reference (:63) for
            # the sorted per-node batch vector, without the host sync a tensor of repeat counts costs (HIP-graph capturable).
            # As a one-hot GEMM, not index_select: the values are the same bit for bit (one non-zero term per output), and the
            # backward is a GEMM with a fixed summation order where index_select's is index_add_ with float atomics -- the rows of
            # one image collide, and the appearance encoder's 24 gradients came out different from run to run (found in round 5
            # by tools/dbg_branch_streams.py on configs[3])
            rows = gnn.batch.repeat_interleave(t_in)
            onehot = (rows.unsqueeze(1) == torch.arange(enc.shape[0], device=enc.device).unsqueeze(0)).to(enc.dtype)
            scene = onehot @ enc.flatten(1)
            fused = self.fuse_appearance_roi(torch.cat([scene, obj], dim=1))
            out["objects_feature"] = torch.cat(fused.unsqueeze(1).chunk(t_in, 0), 1)
        if not self.defer_aux_join:                 # stand-alone use: the caller reads the result on its own stream
            ops.aux_join(out["objects_feature"], lanes=(0,))
        return out
