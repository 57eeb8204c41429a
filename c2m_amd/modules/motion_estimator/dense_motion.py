"""DenseMotionNetwork: sparse user motion -> dense backward flow + occlusion
(reference: src/modules/motion_estimator/dense_motion.py:12-271).

MI355X-first changes with identical results: the objects x frames Python loop of generate_sparse_motion
(affine_grid + grid_sample + 3 torch.where per pair) is ONE raster launch; the ten scatter_add occlusion maps are two
frame-batched exact-order splat pipelines; decisions that the reference takes on device tensors (`inst_id == 0`,
tensor-indexed batch ids) are left on the device."""
import torch
from torch import nn

from ... import ops
from .sparse_motion_estimator import SparseMotionGenerator
from .motion_autoencoder import DenseMotionEncoder, DenseMotionDecoder
from .sparse_encoder import SparseMotionFeatureEncoder
from ..layers.same_block import SameBlockTwoConv2d


class DenseMotionNetwork(nn.Module):
    def __init__(self, train_params, model_params):
        super().__init__()
        self.train_params = train_params
        self.model_params = model_params
        self.defer_aux_join = False      # True (GeneratorFullModel): gt thetas stay on the auxiliary stream until the model's join
        tp, mp, cp = train_params, model_params, model_params["common_params"]
        t_in = tp["num_input_frames"]
        self.num_frames = t_in + tp["num_predicted_frames"]
        self.dense_motion_params = mp["motion_estimator"]
        self.scale_factor = cp["scale_factor"]
        ap = mp["appearance_encoder"]
        self.down_factor = 2 ** ap["num_down_blocks"]
        self.h_appearance_map = int(tp["input_size"][0] / self.down_factor * self.scale_factor)
        self.w_appearance_map = int(tp["input_size"][1] / self.down_factor * self.scale_factor)
        self.h_scene_feature = ap["block_expansion"] * (2 ** ap["pooling_after"])
        self.sparse_motion_estimator = SparseMotionGenerator(
            **self.dense_motion_params["sparse_motion_estimator"], input_scene_features=self.h_scene_feature,
            h_scene_features=self.h_scene_feature, num_predicted_frames=tp["num_predicted_frames"],
            num_input_frames=t_in)
        self.sparse_feature_encoder = SparseMotionFeatureEncoder(self.dense_motion_params["sparse_motion_encoder"])
        enc_params = self.dense_motion_params["dense_motion_encoder"]
        enc_params.update({"scale_factor": self.scale_factor, "input_size": tp["input_size"]})
        zconv_in = min(ap["block_expansion"] * (2 ** ap["num_down_blocks"]), ap["max_expansion"])
        flow_occ = cp["flow_channel"] + cp["occlusion_channel"]
        fg_frame = cp["image_channel"] + cp["seg_channel_fg"] + cp["instance_channel"]
        bg_frame = cp["image_channel"] + cp["seg_channel_bg"]
        self.motion_encoder_fg = DenseMotionEncoder(enc_params, input_channel=fg_frame * t_in + flow_occ + fg_frame,
                                                    output_channel=enc_params["out_channel_fg"])
        self.motion_encoder_bg = DenseMotionEncoder(enc_params, input_channel=bg_frame * t_in + flow_occ + bg_frame,
                                                    output_channel=enc_params["out_channel_bg"])
        dec_params = self.dense_motion_params["dense_motion_decoder"]
        dec_params.update({"num_input_frames": t_in, "num_predicted_frames": tp["num_predicted_frames"],
                           "scale_factor": self.scale_factor, "input_size": tp["input_size"],
                           "sparse_down": self.dense_motion_params["sparse_motion_encoder"]["num_down_blocks"]})
        self.dense_generator_bw = DenseMotionDecoder(dec_params)
        if tp["use_fw_of"]:
            self.dense_generator_fw = DenseMotionDecoder(dec_params)
        self.zconv = SameBlockTwoConv2d(zconv_in + 64, 16 * tp["num_predicted_frames"], 3, 1, 1, padding_mode="reflect")
        self.fc = nn.Linear(enc_params["out_channel_bg"] + enc_params["out_channel_fg"],
                            64 * self.h_appearance_map * self.w_appearance_map)

    def get_parameters(self):
        mods = [self.sparse_feature_encoder, self.motion_encoder_fg, self.motion_encoder_bg, self.dense_generator_bw,
                self.zconv, self.fc]
        if self.train_params["use_fw_of"]:
            mods.append(self.dense_generator_fw)
        return [p for m in mods for p in m.parameters()]

    @staticmethod
    def reparameterize(mu, logvar, eps=None):
        std = torch.exp(0.5 * logvar)
        if eps is None:
            eps = torch.randn_like(std)
        return mu + eps * std

    @staticmethod
    def clip_mask(mask):
        return (mask > 0.5).to(mask.dtype)

    def generate_sparse_motion(self, tracking_gnn, sparse_motion_dict, source_instance, use_gt=False):
        """source_instance [B,1,H,W] float instance ids -> sparse flow / support / occlusion tensors."""
        T = self.train_params["num_predicted_frames"]
        if use_gt:
            thetas = tracking_gnn.targets_theta
        else:
            thetas = torch.stack([sparse_motion_dict[f"theta_{t}"] for t in range(T)], 1)
        ids = tracking_gnn.source_frames_nodes_instance_ids[:, -1]
        bw, fw, binm = ops.sparse_raster(source_instance[:, 0], ids, tracking_gnn.batch, thetas)
        out = {"sparse_motion_bw": bw}
        if self.train_params["use_fw_of"]:
            out["sparse_motion_fw"] = fw
        out["sparse_motion_bin"] = binm
        out["sparse_occ_bw"] = ops.occlusion_splat(fw, want_map=False, want_clip=True)[1]
        out["sparse_occ_fw"] = ops.occlusion_splat(bw, want_map=False, want_clip=True)[1]
        return out, fw

    def _inputs_then_targets(self, x):
        """The two pieces of cat([first t_in frames stacked into channels and repeated over T, the T target frames], 1) (:173-192):
        returned as a list, so that the encoder inputs below are ONE concatenation of views instead of a cat of cats (the
        intermediate copies were 360 MB written and read again per configs[1] step)."""
        t_in, T = self.train_params["num_input_frames"], self.train_params["num_predicted_frames"]
        b, c, _, h, w = x.shape
        src = x[:, :, :t_in].permute(0, 2, 1, 3, 4).reshape(b, t_in * c, 1, h, w).expand(b, t_in * c, T, h, w)
        return [src, x[:, :, t_in:]]

    def _decode(self, app_features, sparse, sparse_fw, z_m, out):
        tp = self.train_params
        T = tp["num_predicted_frames"]
        enc_bw = self.sparse_feature_encoder(sparse["sparse_motion_bw"])
        code = self.zconv(torch.cat([self.fc(z_m).view(-1, 64, self.h_appearance_map, self.w_appearance_map),
                                     app_features["app_encoded"]], 1))
        b, _, h, w = code.shape
        codex = app_features["app_encoded"].unsqueeze(2).expand(-1, -1, T, -1, -1)
        code = code.view(b, T, 16, h, w).permute(0, 2, 1, 3, 4)       # chunk(T, 1) stacked on a new time axis
        z = torch.cat([codex, code], 1)
        if tp["use_fw_of"]:
            enc_fw = self.sparse_feature_encoder(sparse["sparse_motion_fw"])
            dense_fw = self.dense_generator_fw(app_features, enc_fw, sparse["sparse_motion_fw"], sparse["sparse_occ_fw"], z)
        dense_bw = self.dense_generator_bw(app_features, enc_bw, sparse["sparse_motion_bw"], sparse["sparse_occ_bw"], z)
        out.update(sparse)
        out["dense_motion_bw"], out["occlusion_bw"] = dense_bw["dense_motion"], dense_bw["occlusion"]
        if tp["use_fw_of"]:
            out["dense_motion_fw"], out["occlusion_fw"] = dense_fw["dense_motion"], dense_fw["occlusion"]
        return out

    def forward(self, app_features, model_input):
        tp = self.train_params
        t_in = tp["num_input_frames"]
        frames = self._inputs_then_targets(model_input["frames"])
        bg = self._inputs_then_targets(model_input["bg_mask"])
        fg = self._inputs_then_targets(model_input["fg_mask"])
        inst = self._inputs_then_targets(model_input["instance"].to(frames[1].dtype))
        flows = [model_input["target_bw_of"], model_input["target_bw_occ"]]
        bg_out = self.motion_encoder_bg(torch.cat(frames + bg + flows, 1))
        fg_out = self.motion_encoder_fg(torch.cat(frames + fg + inst + flows, 1))
        out = {"mu": torch.cat([bg_out["mu"], fg_out["mu"]], 1),
               "logvar": torch.cat([bg_out["logvar"], fg_out["logvar"]], 1)}
        z_m = self.reparameterize(out["mu"], out["logvar"], model_input.get("eps"))
        # object GNN on the auxiliary stream (ops.aux_branch): with gt thetas its outputs feed only the theta losses, so it runs next
        # to the encoders / decoder / generator and is joined in front of the losses (GeneratorFullModel._forward); with predicted
        # thetas the raster below needs them: joined here
        with ops.aux_branch(*((app_features["objects_feature"], model_input["latent"]) if self.training and torch.is_grad_enabled() else ()), part="gnn"):
            thetas = self.sparse_motion_estimator(model_input["tracking_gnn"], app_features["objects_feature"],
                                                  model_input["latent"], model_input.get("click_index"))
        if not tp["use_gt_training"] or not self.defer_aux_join:
            ops.aux_join(*thetas.values(), lanes=(0,))
        out.update(thetas)
        sparse, fw = self.generate_sparse_motion(model_input["tracking_gnn"], thetas,
                                                 model_input["instance"][:, :, t_in - 1].to(frames[1].dtype),
                                                 tp["use_gt_training"])
        return self._decode(app_features, sparse, fw, z_m, out)

    def inference(self, app_features, model_input):
        tp = self.train_params
        t_in = tp["num_input_frames"]
        thetas = self.sparse_motion_estimator.inference(model_input["tracking_gnn"], model_input["latent_traj"],
                                                        model_input["index_user_guidance"],
                                                        app_features["objects_feature"])
        out = dict(thetas)
        sparse, fw = self.generate_sparse_motion(model_input["tracking_gnn"], thetas,
                                                 model_input["instance"][:, :, t_in - 1].float(), tp["use_gt_eval"])
        out = self._decode(app_features, sparse, fw, model_input["z_m"], out)
        out["index_user_guidance"] = model_input["index_user_guidance"]
        return out
