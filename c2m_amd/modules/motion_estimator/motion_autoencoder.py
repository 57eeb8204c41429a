"""DenseMotionEncoder / DenseMotionDecoder / FlowPredictor / OcclusionPredictor
(reference: src/modules/motion_estimator/motion_autoencoder.py:11-181)."""
import torch
from torch import nn

from ... import ops
from ...utils import resize_flow
from ..layers.same_block import SameBlock3d, SameBlock2d
from ..layers.down_block import DownBlock3d
from ..layers.up_block import UpBlock2d
from ..layers.common import conv_module, fold_time, unfold_time


class DenseMotionEncoder(nn.Module):
    def __init__(self, model_params, input_channel, output_channel):
        super().__init__()
        mp = model_params
        self.scale_factor, self.input_size = mp["scale_factor"], mp["input_size"]
        self.input_channel, self.output_channel = input_channel, output_channel
        self.block_expansion, self.num_down_blocks = mp["block_expansion"], mp["num_down_blocks"]
        self.down_factor = 2 ** self.num_down_blocks
        self.max_expansion, self.padding_mode = mp["max_expansion"], mp["padding_mode"]
        self.h_appearance_map = int(self.input_size[0] / self.down_factor * self.scale_factor)
        self.w_appearance_map = int(self.input_size[1] / self.down_factor * self.scale_factor)
        self.t_s, self.h_s, self.w_s = mp["t_stride"], mp["h_stride"], mp["w_stride"]
        self.t_k, self.h_k, self.w_k = mp["t_kernel"], mp["h_kernel"], mp["w_kernel"]
        self.t_p, self.h_p, self.w_p = mp["t_padding"], mp["h_padding"], mp["w_padding"]
        blocks, width = [], self.input_channel
        for i in range(len(self.w_p)):
            nxt = min(self.max_expansion, self.block_expansion * (2 ** i))
            blocks.append(DownBlock3d(in_features=width, out_features=nxt,
                                      kernel_size=[self.t_k[i], self.h_k[i], self.w_k[i]],
                                      stride=[self.t_s[i], self.h_s[i], self.w_s[i]],
                                      padding=[self.w_p[i]] * 2 + [self.h_p[i]] * 2 + [self.t_p[i]] * 2,
                                      padding_mode=self.padding_mode))
            width = nxt
        self.down_blocks = nn.ModuleList(blocks)
        flat = self.h_appearance_map * self.w_appearance_map * width
        self.fc1 = nn.Linear(flat, self.output_channel)
        self.fc2 = nn.Linear(flat, self.output_channel)

    def forward(self, video):
        x = video
        for blk in self.down_blocks:
            x = blk(x)
        flat = x.reshape(video.shape[0], -1)
        return {"mu": self.fc1(flat), "logvar": self.fc2(flat)}


class FlowPredictor(nn.Module):
    def __init__(self, output_channel=2, input_channel=64):
        super().__init__()
        self.flow_predictor = nn.Sequential(
            SameBlock2d(in_features=input_channel, out_features=32, kernel_size=3, stride=1, padding=1, padding_mode="reflect"),
            nn.ReflectionPad2d(1), nn.Conv2d(32, output_channel, 3, 1, 0))

    def forward(self, x):
        return conv_module(self.flow_predictor[0](x), self.flow_predictor[2], padding=1, padding_mode="reflect")


class OcclusionPredictor(nn.Module):
    def __init__(self, input_channel_features=64, input_conditioning=2):
        super().__init__()
        self.occlusion_predictor = nn.Sequential(
            SameBlock2d(in_features=input_channel_features, out_features=32, kernel_size=3, stride=1, padding=1,
                        padding_mode="reflect"),
            nn.ReflectionPad2d(1), nn.Conv2d(32, 1, 3, 1, 0), nn.Sigmoid())

    def forward(self, x):
        return conv_module(self.occlusion_predictor[0](x), self.occlusion_predictor[2], act="sigmoid", padding=1,
                           padding_mode="reflect")


class DenseMotionDecoder(nn.Module):
    def __init__(self, model_params):
        super().__init__()
        mp = model_params
        self.scale_factor, self.input_size = mp["scale_factor"], mp["input_size"]
        self.input_channel, self.out_channel = mp["in_channel"], mp["out_channel"]
        self.num_input_frames, self.num_predicted_frames = mp["num_input_frames"], mp["num_predicted_frames"]
        self.block_expansion, self.num_up_blocks = mp["block_expansion"], mp["num_up_blocks"]
        self.up_factor = 2 ** self.num_up_blocks
        self.max_expansion, self.padding_mode = mp["max_expansion"], mp["padding_mode"]
        self.num_down_block_sparse_encoder = mp["sparse_down"]
        self.use_feature_resample = mp["use_feature_resample"]
        self.use_appearance_feature = mp["use_appearance_feature"]
        nu, pm = self.num_up_blocks, self.padding_mode

        def width(level):
            return min(self.max_expansion, self.block_expansion * (2 ** level))

        self.first = SameBlock3d(self.input_channel, width(nu), 3, 1, 1, padding_mode=pm)
        ups, fuses, flows, occs = [], [], [], []
        for i in range(nu):
            cin = width(nu - i)
            if i > 0 and self.use_appearance_feature:
                cin *= self.num_input_frames + 1
            cout = width(nu - i - 1)
            ups.append(UpBlock2d(cin, cout, padding_mode=pm))
            flows.append(FlowPredictor(output_channel=2, input_channel=cout))           # built, never called (:93-95)
            occs.append(OcclusionPredictor(input_channel_features=cout, input_conditioning=0))
            if i >= nu - self.num_down_block_sparse_encoder:
                fuses.append(SameBlock3d(cout * 2, cout, 3, 1, 1, padding_mode=pm))
        self.up_blocks = nn.ModuleList(ups)
        self.fuse_convs = nn.ModuleList(fuses)
        self.flow_predictors = nn.ModuleList(flows)
        self.occlusion_predictors = nn.ModuleList(occs)
        self.final_up_block = UpBlock2d(cout, self.out_channel, padding_mode=pm)
        self.final_fuse = SameBlock3d(cout + 2, cout, 3, 1, 1, padding_mode=pm)
        self.flow = FlowPredictor(output_channel=2, input_channel=cout)
        self.occlusion = OcclusionPredictor(input_channel_features=cout, input_conditioning=0)

    @staticmethod
    def _match(x5, hw):
        if list(x5.shape[-2:]) == list(hw):
            return x5
        raise NotImplementedError("feature-map size mismatch in DenseMotionDecoder (needs a differentiable resize)")

    def forward(self, appearance_features, sparse_features, sparse_motion, sparse_occlusion, z):
        T, nu = self.num_predicted_frames, self.num_up_blocks
        out = self.first(z)
        flat_motion = flat_occ = None
        fuse_i = 0
        for i, up in enumerate(self.up_blocks):
            inp = out
            if i > 0 and self.use_appearance_feature:
                feat = appearance_features[f"enco{nu - i}"]
                b, c, h, w = feat.shape
                rep = feat.unsqueeze(0).expand(T, b, c, h, w).reshape(T * b, c, h, w)   # frame-major repeat
                if self.use_feature_resample:
                    if flat_motion is None:
                        flat_motion, flat_occ = fold_time(sparse_motion), fold_time(sparse_occlusion)
                    motion = resize_flow(flat_motion, [h, w])
                    occ = ops.resize_bilinear(flat_occ, (h, w), align_corners=False)
                    rep = ops.flow_warp(rep, motion, occ)          # resample(app, flow) * occlusion, one kernel
                out = self._match(out, (h, w))
                inp = torch.cat([out, unfold_time(rep, T)], 1)
            out = up(inp)
            if i >= nu - self.num_down_block_sparse_encoder:
                sf = sparse_features[f"enco_sparse_{nu - i - 1}"]
                out = self._match(out, sf.shape[-2:])
                out = self.fuse_convs[fuse_i](torch.cat([out, sf], 1))
                fuse_i += 1
        out = self.final_up_block(out)
        # the rastered sparse motion carries no gradient: the data gradient of final_fuse is only needed for `out`'s channels
        keep = None if sparse_motion.requires_grad else out.shape[1]
        out = fold_time(self.final_fuse(torch.cat([out, sparse_motion], dim=1), dgrad_channels=keep))
        return {"dense_motion": unfold_time(self.flow(out), T), "occlusion": unfold_time(self.occlusion(out), T)}
