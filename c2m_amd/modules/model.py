"""GeneratorFullModel: the whole training-step graph (reference: src/modules/model.py:30-324).

Same constructor, attributes (optimizer, optimizer_gnn, scheduler_*, d_optimizer_*), forward/inference signatures,
output-dict keys and state_dict keys as the reference, so src/train.py + Trainer.update_model drive it unchanged.
Randomness can be injected through data_batch["rng"] = {latent_traj, eps, click_index} (parity runs / graph capture);
otherwise it is drawn like the reference does (latent on the host, eps on the device, click index with NumPy)."""
import functools

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

from .. import ops
from ..optim import Adam
from ..utils import utils as U
from ..losses import losses
from .appearance_encoder.appearance_encoder import AppearanceEncoder
from .motion_estimator.dense_motion import DenseMotionNetwork
from .generator.generator import OcclusionAwareGenerator
from .discriminator import discriminator
from .layers.common import fold_time, unfold_time, deferred_batch_counters


def get_norm_layer(norm_type='instance'):
    if norm_type == 'batch':
        return functools.partial(nn.BatchNorm2d, affine=True)
    if norm_type == 'instance':
        return functools.partial(nn.InstanceNorm2d, affine=False, track_running_stats=True)
    raise NotImplementedError('normalization layer [%s] is not found' % norm_type)


def _stack_time(x):
    """[B,C,T,H,W] -> [B,T*C,H,W] (cat(unbind(x, 2), 1))."""
    b, c, t, h, w = x.shape
    return x.permute(0, 2, 1, 3, 4).reshape(b, t * c, h, w)


def _linear_input_cast(module, args):
    x = args[0]
    return (x.to(module.weight.dtype),) if x.dtype != module.weight.dtype else None


def _cast_linear_inputs(root):
    """bf16 data path (ops.set_conv_precision("bf16")): convolution outputs are bf16 while the few nn.Linear layers of the
    path (VAE heads, RoI regressors, GNN) keep fp32 weights and run on torch's GEMM; their inputs are widened at the boundary
    ([B, 4096]-sized vectors).  A no-op in fp32 mode."""
    for m in root.modules():
        if isinstance(m, nn.Linear):
            m.register_forward_pre_hook(_linear_input_cast)


class GeneratorFullModel(nn.Module):
    def __init__(self, train_params=None, model_params=None, is_inference=False, dataset="cityscape"):
        super().__init__()
        self.train_params = train_params
        self.model_params = model_params
        tp, mp = train_params, model_params
        mp["generator"].setdefault("use_spade", True)   # absent from the shipped YAML
        self.num_frames = tp["num_input_frames"] + tp["num_predicted_frames"]
        self.appearance_encoder = AppearanceEncoder(tp, **mp["appearance_encoder"], **mp["common_params"])
        self.motion_encoder = DenseMotionNetwork(tp, mp)
        # the object branch (RoI head + GNN) stays on its auxiliary stream across the two modules; _forward joins it (ops.aux_branch)
        self.appearance_encoder.defer_aux_join = self.motion_encoder.defer_aux_join = True
        self.criterionGAN = discriminator.GANLoss()
        self.criterionFeat = torch.nn.L1Loss()
        self.generator = OcclusionAwareGenerator(mp["generator"], mp["flow_embedder"],
                                                 input_channel=mp["common_params"]["image_channel"], dataset=dataset)
        _cast_linear_inputs(self)
        if is_inference:
            return
        self.objective_func = losses.TrainingLosses(tp, mp)
        # torch.optim.Adam state layout / schedulers, stepped by the HIP multi-tensor kernel (c2m_amd/optim.py)
        adam = functools.partial(Adam, betas=(tp["beta1"], tp["beta2"]), eps=float(tp["eps"]))
        milestones = list(range(tp["milestone_start"], tp["milestone_end"], tp["milestone_every"]))
        sched = functools.partial(torch.optim.lr_scheduler.MultiStepLR, milestones=milestones)
        self.model_parameters = list(self.appearance_encoder.parameters()) + \
            list(self.motion_encoder.get_parameters()) + list(self.generator.parameters())
        self.optimizer = adam(self.model_parameters, lr=tp["lr_rate_g"])
        self.optimizer_gnn = adam(list(self.motion_encoder.sparse_motion_estimator.parameters()), lr=tp["lr_rate_gnn"])
        self.scheduler_g = sched(self.optimizer, gamma=tp["gamma_g"])
        self.scheduler_gnn = sched(self.optimizer_gnn, gamma=tp["gamma_gnn"])
        dp = mp["discriminator"]
        if tp["use_image_discriminator"]:
            self.netD_image = discriminator.define_d(dp["in_channel"], dp["ndf"], dp["n_layers_D"], dp["num_D"],
                                                     dp["padding_mode"])
            self.d_optimizer_image = adam(list(self.netD_image.parameters()), lr=tp["lr_rate_d"])
            self.scheduler_d_image = sched(self.d_optimizer_image, gamma=tp["gamma_d"])
        if tp["use_video_discriminator"]:
            self.netD_video = discriminator.define_d(self.num_frames * dp["in_channel"], dp["ndf"], dp["n_layers_D"],
                                                     dp["num_D"], dp["padding_mode"])
            self.d_optimizer_video = adam(list(self.netD_video.parameters()), lr=tp["lr_rate_d"])
            self.scheduler_d_video = sched(self.d_optimizer_video, gamma=tp["gamma_d"])

    # ------------------------------------------------------------------------------------------ discriminator terms
    def compute_loss_d(self, net_d, gt, fake, dis_type="image"):
        pred_real = net_d.forward(gt)
        pred_fake = net_d.forward(fake.detach())
        key = 'prediction_map_%s' % 0
        loss_d_real = self.criterionGAN(pred_real[key], True)
        loss_d_fake = self.criterionGAN(pred_fake[key], False)
        pred_fake = net_d.forward(fake)          # second pass on the attached fake: G loss also reaches D params
        loss_g_gan, loss_g_gan_feat = self.gan_and_fm_loss(pred_real, pred_fake, dis_type)
        return loss_d_real, loss_d_fake, loss_g_gan, loss_g_gan_feat

    def gan_and_fm_loss(self, pred_real, pred_fake, dis_type):
        loss_g_gan = self.criterionGAN(pred_fake['prediction_map_%s' % 0], True)
        loss_fm = 0
        if self.train_params["loss_weights"][f"feature_matching_{dis_type}"] > 0:
            for a, b in zip(pred_real['feature_maps_%s' % 0], pred_fake['feature_maps_%s' % 0]):
                loss_fm = loss_fm + ops.l1_mean(b, a.detach())
        return loss_g_gan, loss_fm

    # ------------------------------------------------------------------------------------------ shared front end
    def _resize_inputs(self, get):
        sf = self.model_params["common_params"]["scale_factor"]
        r = U.resize_video
        return dict(frames=r(get("video"), sf, mode="bilinear"), bg_mask=r(get("bg_mask"), sf, mode="nearest"),
                    fg_mask=r(get("fg_mask"), sf, mode="nearest"),
                    instance=r(get("instance_mask").float(), sf, mode="nearest").int(),
                    input_of=r(get("input_of"), sf, mode="bilinear", is_flow=True),
                    input_occ=r(get("input_occ"), sf, mode="bilinear"))

    def _encoder_input(self, v):
        t_in = self.train_params["num_input_frames"]
        seg = torch.cat([v["bg_mask"][:, :, :t_in], v["fg_mask"][:, :, :t_in]], 1)
        parts = [_stack_time(v["frames"][:, :, :t_in]), _stack_time(seg),
                 _stack_time(v["instance"][:, :, :t_in]).to(v["frames"].dtype)]
        if v["input_of"] is not None:
            parts += [_stack_time(v["input_of"][:, :, :t_in]), _stack_time(v["input_occ"][:, :, :t_in])]
        return torch.cat(parts, 1)

    def _draw_latent(self, gnn, device):
        tp = self.train_params
        z_dim = self.model_params["motion_estimator"]["sparse_motion_estimator"]["z_dim"]
        return torch.FloatTensor(gnn.x.shape[0], tp["num_predicted_frames"], z_dim).normal_(0, 1).to(device)

    def _generate(self, v, out):
        """Image generator on the T folded frames + the two sparse-flow visualisations (model.py:195-211)."""
        tp = self.train_params
        t_in, T = tp["num_input_frames"], tp["num_predicted_frames"]
        last = v["frames"][:, :, t_in - 1]
        b, c, h, w = last.shape
        rep = last.unsqueeze(0).expand(T, b, c, h, w).reshape(T * b, c, h, w)
        gen = self.generator(rep, fold_time(out["dense_motion_bw"]), fold_time(out["occlusion_bw"]))
        out["generated"] = unfold_time(gen, T)
        with torch.no_grad():
            sparse = fold_time(out["sparse_motion_bw"])
            out["generated_sparse"] = unfold_time(ops.flow_warp(rep, sparse), T)
            out["generated_sparse_occ"] = unfold_time(ops.flow_warp(rep, sparse, fold_time(out["sparse_occ_bw"])), T)
        return out

    # ------------------------------------------------------------------------------------------ training forward
    def forward(self, data_batch):
        with deferred_batch_counters():          # the BatchNorm step counters: one multi-tensor add per forward
            return self._forward(data_batch)

    def _forward(self, data_batch):
        tp = self.train_params
        t_in = tp["num_input_frames"]
        v = self._resize_inputs(data_batch.get)
        sf = self.model_params["common_params"]["scale_factor"]
        target_bw_of = U.resize_video(data_batch.get("target_bw_of"), sf, mode="bilinear", is_flow=True)
        target_bw_occ = U.resize_video(data_batch.get("target_bw_occ"), sf, mode="bilinear")
        target_fw_of = U.resize_video(data_batch.get("target_fw_of"), sf, mode="bilinear", is_flow=True)
        target_fw_occ = U.resize_video(data_batch.get("target_fw_occ"), sf, mode="bilinear")
        pl = getattr(self.objective_func, "perceptual_loss", None)
        if pl is not None:
            pl.prefetch(v["frames"][:, :, t_in:])          # ground-truth VGG features: auxiliary stream, joined before the losses
        gnn = data_batch["tracking_gnn"]
        rng = data_batch.get("rng") or {}
        latent = rng["latent_traj"] if "latent_traj" in rng else self._draw_latent(gnn, gnn.x.device)
        app = self.appearance_encoder({"first_frame": self._encoder_input(v), "tracking_gnn": gnn})
        motion_input = dict(frames=v["frames"], bg_mask=v["bg_mask"], fg_mask=v["fg_mask"], instance=v["instance"],
                            input_of=v["input_of"], input_occ=v["input_occ"], target_bw_of=target_bw_of,
                            target_bw_occ=target_bw_occ, target_fw_of=target_fw_of, target_fw_occ=target_fw_occ,
                            tracking_gnn=gnn, latent=latent, eps=rng.get("eps"), click_index=rng.get("click_index"))
        out = {}
        out.update(self.motion_encoder(app, motion_input))
        out = self._generate(v, out)
        # the object branch and the ground-truth VGG pass (ops.aux_branch) meet the main stream
        ops.aux_join(*[t for k, t in out.items() if k.startswith("theta_")], *(pl.prefetched_tensors() if pl is not None else []))
        loss_dict = self.objective_func(data_batch["video"], v["frames"], target_bw_of, target_fw_of, target_bw_occ,
                                        target_fw_occ, out, gnn)
        loss_d_image, loss_d_video = {}, {}
        if tp["use_image_discriminator"]:
            d_real, d_fake, g_gan, g_fm = self.compute_loss_d(
                self.netD_image, fold_time(data_batch["video"][:, :, t_in:]), fold_time(out["generated"]), "image")
            loss_dict["g_gan_image"], loss_dict["feature_matching_image"] = g_gan, g_fm
            loss_d_image = {"d_real": d_real, "d_fake": d_fake}
        if tp["use_video_discriminator"]:
            fake = torch.cat([_stack_time(v["frames"][:, :, :t_in]), _stack_time(out["generated"])], dim=1)
            d_real, d_fake, g_gan, g_fm = self.compute_loss_d(self.netD_video, _stack_time(v["frames"]), fake, "video")
            loss_dict["g_gan_video"], loss_dict["feature_matching_video"] = g_gan, g_fm
            loss_d_video = {"d_real": d_real, "d_fake": d_fake}
        return out, loss_dict, loss_d_image, loss_d_video

    # ------------------------------------------------------------------------------------------ inference
    def inference(self, video, bg_mask, fg_mask, instance_mask, input_of, input_occ, tracking_gnn=None,
                  index_user_guidance=None, z_m=None):
        tp = self.train_params
        latent = self._draw_latent(tracking_gnn, video.device)
        self.motion_encoder.sparse_motion_estimator.eval()
        if index_user_guidance is None:
            index_user_guidance = self.motion_encoder.sparse_motion_estimator.draw_click_index(
                tracking_gnn.num_real_nodes, video.device)
        src = dict(video=video, bg_mask=bg_mask, fg_mask=fg_mask, instance_mask=instance_mask, input_of=input_of,
                   input_occ=input_occ)
        v = self._resize_inputs(src.get)
        app = self.appearance_encoder({"first_frame": self._encoder_input(v), "tracking_gnn": tracking_gnn})
        ops.aux_join(app["objects_feature"])        # (a no-op unless inference runs on a model in training mode with grad enabled)
        out = {}
        out.update(self.motion_encoder.inference(app, dict(instance=v["instance"], latent_traj=latent, z_m=z_m,
                                                           index_user_guidance=index_user_guidance,
                                                           tracking_gnn=tracking_gnn)))
        return self._generate(v, out)
