"""TEST INFRASTRUCTURE -- CPU restatement of the C2M generator train-step hot path.  NOT product code.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module;
the product (c2m_amd/) never does.  It is the checker, never the thing measured or shipped.

What it is: a *functional* restatement (no nn.Module tree) of the reference algorithm, driven by
a flat `state` dict that uses the reference's own state_dict key names, written against plain
torch CPU ops.  Each function cites the reference file:line it follows (paths relative to
/root/reference/src).  Parity status: PINNED -- tests/test_oracle_golden.py checks it against golden
vectors captured from the live reference by oracle/capture_golden.py (losses, outputs, gradients,
BatchNorm buffers), except roi_align / GATv2Conv / VGG-19 weights which are third-party and
unpinned (oracle/thirdparty.py).

Randomness is injected: `rng = {latent_traj, eps, click_index}` (model.py:157-160,
dense_motion.py:88-92, sparse_motion_estimator.py:46-51).
"""
import math

import torch
import torch.nn.functional as F

from . import thirdparty

LRELU = 0.2
BN_EPS = 1e-5
BN_MOM = 0.1


# --------------------------------------------------------------------------------------------
# small tensor helpers
# --------------------------------------------------------------------------------------------
def fold_time(x):
    """[B,C,T,H,W] -> [T*B,C,H,W] (frame-major), the `torch.cat(torch.unbind(x, 2), 0)` idiom (e.g. up_block.py:20)."""
    return torch.cat(torch.unbind(x, 2), 0)


def unfold_time(x, t):
    """[T*B,C,H,W] -> [B,C,T,H,W]; `torch.cat(x.unsqueeze(2).chunk(t, 0), 2)` (up_block.py:25)."""
    return torch.cat(x.unsqueeze(2).chunk(t, 0), 2)


def stack_time_into_channels(x):
    """[B,C,T,H,W] -> [B,T*C,H,W]; `torch.cat(torch.unbind(x, 2), 1)` (model.py:163-166)."""
    return torch.cat(torch.unbind(x, 2), 1)


def base_grid(n, h, w):
    """utils/ops.py:196-202 get_grid: linspace(-1,1) grid in the align_corners=True convention, [n,2,h,w]."""
    g = torch.zeros(n, 2, h, w)
    lx = torch.linspace(-1, 1, w) if w > 1 else torch.tensor([-1.0])
    ly = torch.linspace(-1, 1, h) if h > 1 else torch.tensor([-1.0])
    g[:, 0] = torch.ger(torch.ones(h), lx)
    g[:, 1] = torch.ger(ly, torch.ones(w))
    return g


def resample(image, flow):
    """utils/ops.py:187-193: pixel-unit backward warp; align-True grid sampled with align_corners=False, border pad."""
    n, _, h, w = image.shape
    nf = torch.cat([flow[:, 0:1] / ((w - 1.0) / 2.0), flow[:, 1:2] / ((h - 1.0) / 2.0)], 1)
    grid = (base_grid(n, h, w) + nf).permute(0, 2, 3, 1)
    return F.grid_sample(image, grid, mode="bilinear", padding_mode="border", align_corners=False)


def resize_flow(flow, new_hw):
    """utils/utils.py:346-354: bilinear align_corners=True resize, then rescale magnitudes."""
    h, w = flow.shape[-2:]
    nh, nw = new_hw
    out = F.interpolate(flow, (nh, nw), mode="bilinear", align_corners=True)
    out = torch.cat([out[:, 0:1] / (w / float(nw)), out[:, 1:2] / (h / float(nh))], 1)
    return out


def corresponding_map(coords):
    """utils/ops.py:205-251: forward splat of bilinear weights with scatter_add (order = the reference's cat order)."""
    b, _, h, w = coords.shape
    x = coords[:, 0].reshape(b, -1)
    y = coords[:, 1].reshape(b, -1)
    xf_raw, yf_raw = torch.floor(x), torch.floor(y)
    xc_raw, yc_raw = xf_raw + 1, yf_raw + 1
    xf, yf = xf_raw.clamp(0, w - 1), yf_raw.clamp(0, h - 1)
    xc, yc = xc_raw.clamp(0, w - 1), yc_raw.clamp(0, h - 1)
    bad_xc, bad_yc, bad_xf, bad_yf = xc_raw != xc, yc_raw != yc, xf_raw != xf, yf_raw != yf
    invalid = torch.cat([bad_xc | bad_yc, bad_xc | bad_yf, bad_xf | bad_yc, bad_xf | bad_yf], 1)
    idx = torch.cat([xc + yc * w, xc + yf * w, xf + yc * w, xf + yf * w], 1).long()
    wx_c, wx_f = 1 - torch.abs(x - xc), 1 - torch.abs(x - xf)
    wy_c, wy_f = 1 - torch.abs(y - yc), 1 - torch.abs(y - yf)
    val = torch.cat([wx_c * wy_c, wx_c * wy_f, wx_f * wy_c, wx_f * wy_f], 1)
    val = torch.where(invalid, torch.zeros_like(val), val)
    acc = torch.zeros(b, h * w, dtype=coords.dtype)
    acc.scatter_add_(1, idx, val)
    return acc.view(b, 1, h, w)


def occlusion_map(flow):
    """utils/ops.py:254-275 mesh_grid + get_occlusion_map (no grad)."""
    b, _, h, w = flow.shape
    xs = torch.arange(0, w).repeat(b, h, 1)
    ys = torch.arange(0, h).repeat(b, w, 1).transpose(1, 2)
    grid = torch.stack([xs, ys], 1).type_as(flow)
    with torch.no_grad():
        cm = corresponding_map(grid + flow)
    return cm.clamp(min=0.0, max=1.0)


def clip_mask(m):
    """dense_motion.py:155-160."""
    return torch.where(m > 0.5, torch.ones_like(m), torch.zeros_like(m))


# --------------------------------------------------------------------------------------------
# state access + layers
# --------------------------------------------------------------------------------------------
class State:
    """Flat name->tensor view with the reference's state_dict keys.  Trainable leaves get requires_grad."""

    def __init__(self, tensors, trainable=True, frozen_prefixes=("objective_func.perceptual_loss.vgg19.",)):
        self.t = {}
        for k, v in tensors.items():
            v = v.detach().clone()
            is_buf = k.endswith(("running_mean", "running_var", "num_batches_tracked", "weight_u", "weight_v"))
            if trainable and v.is_floating_point() and not is_buf and not k.startswith(tuple(frozen_prefixes)):
                v.requires_grad_(True)
            self.t[k] = v

    def __getitem__(self, k):
        return self.t[k.lstrip(".")]

    def has(self, k):
        return k.lstrip(".") in self.t

    def grads(self):
        return {k: v.grad for k, v in self.t.items() if v.requires_grad and v.grad is not None}


def _pad_mode(mode):
    return {"zeros": "constant", "reflect": "reflect", "replicate": "replicate"}[mode]


def conv2d(S, p, x, stride=1, padding=0, padding_mode="zeros"):
    w = S[p + ".weight"]
    b = S[p + ".bias"] if S.has(p + ".bias") else None
    if padding and padding_mode != "zeros":
        x = F.pad(x, (padding,) * 4, mode=_pad_mode(padding_mode))
        padding = 0
    return F.conv2d(x, w, b, stride=stride, padding=padding)


def conv3d_prepadded(S, p, x, stride, pad6, padding_mode):
    """down_block.py:35-46 / same_block.py:55-66: explicit ReflectionPad3d then Conv3d(padding=0)."""
    if any(pad6):
        x = F.pad(x, tuple(pad6), mode=_pad_mode(padding_mode))
    return F.conv3d(x, S[p + ".weight"], S[p + ".bias"], stride=stride)


def batch_norm(S, p, x, training=True):
    """nn.BatchNorm{1,2,3}d(affine=True) in train mode, incl. the running-stat side effects."""
    if training:
        S.t[(p + ".num_batches_tracked").lstrip(".")] += 1
    return F.batch_norm(x, S[p + ".running_mean"], S[p + ".running_var"], S[p + ".weight"], S[p + ".bias"],
                        training, BN_MOM, BN_EPS)


def instance_norm(S, p, x, affine=True):
    w = S[p + ".weight"] if affine else None
    b = S[p + ".bias"] if affine else None
    return F.instance_norm(x, None, None, w, b, True, BN_MOM, BN_EPS)


def down_block2d(S, p, x, padding_mode, training=True):
    """layers/down_block.py:5-23 with k4 s2 p1 (all call sites)."""
    y = conv2d(S, p + ".conv", x, stride=2, padding=1, padding_mode=padding_mode)
    return F.leaky_relu(batch_norm(S, p + ".norm", y, training), LRELU)


def same_block2d(S, p, x, k, padding_mode, use_norm=True):
    """layers/same_block.py:5-23."""
    y = conv2d(S, p + ".conv", x, stride=1, padding=k // 2 if k > 1 else 0, padding_mode=padding_mode)
    if use_norm:
        y = instance_norm(S, p + ".norm", y)
    return F.leaky_relu(y, LRELU)


def same_block_two_conv2d(S, p, x, padding_mode):
    """layers/same_block.py:26-47: conv -> IN(affine) -> lrelu -> conv2 (k3 s1 p1 at its only call site)."""
    y = conv2d(S, p + ".conv", x, 1, 1, padding_mode)
    y = F.leaky_relu(instance_norm(S, p + ".norm", y), LRELU)
    return conv2d(S, p + ".conv2", y, 1, 1, padding_mode)


def block3d(S, p, x, stride, pad3, padding_mode, training=True):
    """DownBlock3d (down_block.py:26-48) and SameBlock3d (same_block.py:50-68): pad -> conv3d -> BN3d -> lrelu.
    pad3 = (pt, ph, pw); ReflectionPad3d order is (w,w,h,h,t,t)."""
    pt, ph, pw = pad3
    y = conv3d_prepadded(S, p + ".conv", x, stride, (pw, pw, ph, ph, pt, pt), padding_mode)
    return F.leaky_relu(batch_norm(S, p + ".norm", y, training), LRELU)


def up_block2d(S, p, x4, padding_mode, training=True):
    """layers/up_block.py:5-27 on an already time-folded [T*B,C,H,W] input."""
    y = F.interpolate(x4, scale_factor=2, mode="bilinear")
    y = conv2d(S, p + ".main.1", y, 1, 1, padding_mode)
    return F.leaky_relu(batch_norm(S, p + ".main.2", y, training), LRELU)


def spade_norm(S, p, x, cond):
    """layers/spade_block.py:58-77 (single conditional input, 128 hidden filters, nearest resize)."""
    out = F.instance_norm(x, None, None, None, None, True, BN_MOM, BN_EPS)
    label = F.interpolate(cond, size=x.shape[2:], mode="nearest")
    hid = F.leaky_relu(conv2d(S, p + ".mlps.0.0.conv", label, 1, 1, "reflect"), LRELU)
    gb = conv2d(S, p + ".mlps.0.1", hid, 1, 1, "reflect")
    gamma, beta = gb.chunk(2, dim=1)
    return out * (1 + gamma) + beta


def residual_block(S, p, x, training=True):
    """layers/residual_block.py:6-31."""
    y = F.relu(batch_norm(S, p + ".norm1", x, training))
    y = conv2d(S, p + ".conv1", F.pad(y, (1, 1, 1, 1), mode="reflect"))
    y = F.relu(batch_norm(S, p + ".norm2", y, training))
    y = conv2d(S, p + ".conv2", F.pad(y, (1, 1, 1, 1), mode="reflect"))
    return y + x


def residual_spade_block(S, p, x, cond):
    """layers/residual_block.py:34-71."""
    d = F.leaky_relu(spade_norm(S, p + ".norm1", x, cond), LRELU)
    d = conv2d(S, p + ".conv1", F.pad(d, (1, 1, 1, 1), mode="reflect"))
    d = F.leaky_relu(spade_norm(S, p + ".norm2", d, cond), LRELU)
    d = conv2d(S, p + ".conv2", F.pad(d, (1, 1, 1, 1), mode="reflect"))
    if S.has(p + ".conv_s.weight"):
        s = F.leaky_relu(spade_norm(S, p + ".norm_s", x, cond), LRELU)
        return d + F.conv2d(s, S[p + ".conv_s.weight"])
    return d


def linear(S, p, x):
    return F.linear(x, S[p + ".weight"], S[p + ".bias"])


# --------------------------------------------------------------------------------------------
# sub-networks
# --------------------------------------------------------------------------------------------
def appearance_encoder(S, cfg, first_frame, gnn, training=True, p="appearance_encoder"):
    """appearance_encoder/appearance_encoder.py:54-78 (quirks of :57-62 and :72-76 kept verbatim)."""
    tp, ap = cfg["train_params"], cfg["model_params"]["appearance_encoder"]
    t_in, nd = tp["num_input_frames"], ap["num_down_blocks"]
    out = {}
    boxes = torch.cat([gnn.batch.unsqueeze(1).repeat_interleave(t_in, dim=0),
                       torch.cat(torch.unbind(gnn.source_frames_nodes_roi_padded, dim=1), dim=0)], dim=1)
    x = first_frame
    for i in range(nd):
        x = down_block2d(S, f"{p}.down_blocks.{i}", x, ap["padding_mode"], training)
        out["app_encoded" if i == nd - 1 else f"enco{i}"] = x
    scale_factor = cfg["model_params"]["common_params"]["scale_factor"]
    spatial_scale = (1 / scale_factor) * 2 ** ap["pooling_after"]
    feat = torch.cat(out[f"enco{ap['pooling_after'] - 1}"].chunk(t_in, 1), dim=0)
    obj = thirdparty.roi_align(feat, boxes, ap["pool_size"], spatial_scale=1 / spatial_scale)
    obj = F.leaky_relu(conv2d(S, f"{p}.roi_align_blocks.0.conv", obj), LRELU)  # SameBlock2d(use_norm=False), k7 p0
    obj = linear(S, f"{p}.roi_align_blocks.2", obj.flatten(1))
    obj = linear(S, f"{p}.roi_align_regressor", obj)
    rep = torch.repeat_interleave(out["app_encoded"].flatten(1), gnn.num_real_nodes * t_in, dim=0)
    fused = linear(S, f"{p}.fuse_appearance_roi", torch.cat([rep, obj], dim=1))
    out["objects_feature"] = torch.cat(fused.unsqueeze(1).chunk(t_in, 0), 1)
    return out


def dense_motion_encoder(S, p, cfg, video, training=True):
    """motion_estimator/motion_autoencoder.py:11-59."""
    mp = cfg["model_params"]["motion_estimator"]["dense_motion_encoder"]
    x = video
    for i in range(len(mp["w_padding"])):
        x = block3d(S, f"{p}.down_blocks.{i}", x,
                    (mp["t_stride"][i], mp["h_stride"][i], mp["w_stride"][i]),
                    (mp["t_padding"][i], mp["h_padding"][i], mp["w_padding"][i]), mp["padding_mode"], training)
    flat = x.reshape(video.shape[0], -1)
    return linear(S, p + ".fc1", flat), linear(S, p + ".fc2", flat)


def sparse_motion_generator(S, cfg, gnn, scene_features, latent, click_index, training=True,
                            p="motion_encoder.sparse_motion_estimator"):
    """motion_estimator/sparse_motion_estimator.py:39-61 (generator) and :126-141 (decoder)."""
    T = cfg["train_params"]["num_predicted_frames"]
    heads = 4  # sparse_motion_estimator.py:91 default num_head
    x_n, theta_gt, edge_index = gnn.x, gnn.targets_theta, gnn.edge_index
    u = torch.zeros(gnn.num_nodes)
    u[click_index] = 1
    u = u.unsqueeze(1)

    def mlp2(prefix, v):
        return linear(S, prefix + ".2", F.leaky_relu(linear(S, prefix + ".0", v), LRELU))

    x_map = mlp2(p + ".x_encoder", x_n)
    y_n = mlp2(p + ".y_encoder", theta_gt)
    h = torch.cat(torch.unbind(torch.cat([x_map, scene_features], dim=2), 1), 1)
    q = p + ".encode_scene_features"
    h = F.leaky_relu(batch_norm(S, q + ".1", linear(S, q + ".0", h), training), LRELU)
    h = F.leaky_relu(batch_norm(S, q + ".4", linear(S, q + ".3", h), training), LRELU)
    h = linear(S, q + ".6", h)
    d = p + ".decoder"
    # :127-128 -- y_n is overwritten in place but never read afterwards (dead compute, keeps linear_z in the graph)
    for t in range(T):
        y_n[:, t] = mlp2(d + ".linear_z", latent[:, t]) * (1 - u) + y_n[:, t] * u
    out = {}
    x = h
    for t in range(T):
        c = f"{d}.conv_time_steps.{t}"
        x = thirdparty.gatv2_conv(x, edge_index, S[c + ".lin_l.weight"], S[c + ".lin_l.bias"],
                                  S[c + ".lin_r.weight"], S[c + ".lin_r.bias"], S[c + ".att"], S[c + ".bias"], heads)
        loc = mlp2(f"{d}.loc_time_steps.{t}", x)
        out[f"theta_{t}"] = loc * (1 - u) + theta_gt[:, t] * u
    return out


def warp_object(theta, mask, grid_base):
    """dense_motion.py:162-168: affine_grid (align False) minus the align-True base grid -> pixel flow; mask warp."""
    grid = F.affine_grid(theta.unsqueeze(0), mask.shape, align_corners=False)
    _, _, h, w = mask.shape
    flow = grid - grid_base
    flow = torch.cat([flow[..., 0:1] * ((w - 1.0) / 2.0), flow[..., 1:2] * ((h - 1.0) / 2.0)], dim=-1)
    warped = F.grid_sample(mask, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    return warped, flow.permute(0, 3, 1, 2)


def generate_sparse_motion(cfg, gnn, thetas, source_instance, use_gt):
    """dense_motion.py:94-153."""
    T = cfg["train_params"]["num_predicted_frames"]
    b, _, h, w = source_instance.shape
    bw = torch.zeros(b, 2, T, h, w)
    fw = torch.zeros(b, 2, T, h, w)
    binm = torch.zeros(b, 1, T, h, w)
    gb = base_grid(1, h, w).permute(0, 2, 3, 1)
    ids = gnn.source_frames_nodes_instance_ids[:, -1].long()
    for n in range(ids.shape[0]):
        inst_id, bi = int(ids[n]), int(gnn.batch[n])
        if inst_id == 0:
            continue
        mask = (source_instance[bi] == inst_id).float()
        for t in range(T):
            theta = gnn.targets_theta[n][t] if use_gt else thetas[f"theta_{t}"][n]
            warped, flow = warp_object(theta.view(2, 3), mask.unsqueeze(0), gb)
            bw[bi, :, t] = torch.where(warped == 1, flow, bw[bi, :, t])
            fw[bi, :, t] = torch.where(mask == 1, flow * -1, fw[bi, :, t])
            binm[bi, :, t] = torch.where(warped == 1, warped, binm[bi, :, t])
    out = {"sparse_motion_bw": bw.detach()}
    if cfg["train_params"].get("use_fw_of", False):                                  # :144-145 (key order of the reference's dict)
        out["sparse_motion_fw"] = fw.detach()
    out.update({"sparse_motion_bin": binm, "_sparse_motion_fw": fw.detach()})
    out["sparse_occ_bw"] = torch.stack([clip_mask(occlusion_map(fw[:, :, i])) for i in range(T)], 2)
    out["sparse_occ_fw"] = torch.stack([clip_mask(occlusion_map(bw[:, :, i])) for i in range(T)], 2)
    return out


def sparse_feature_encoder(S, cfg, sparse_motion, training=True, p="motion_encoder.sparse_feature_encoder"):
    """motion_estimator/sparse_encoder.py:6-28."""
    mp = cfg["model_params"]["motion_estimator"]["sparse_motion_encoder"]
    out, x = {}, sparse_motion
    for i in range(mp["num_down_blocks"]):
        x = block3d(S, f"{p}.down_blocks.{i}", x, (1, 2, 2), (1, 1, 1), mp["padding_mode"], training)
        out[f"enco_sparse_{i}"] = x
    return out


def predictor_head(S, p, x, key, sigmoid):
    """motion_autoencoder.py:152-181 FlowPredictor / OcclusionPredictor."""
    y = same_block2d(S, f"{p}.{key}.0", x, 3, "reflect")
    y = conv2d(S, f"{p}.{key}.2", F.pad(y, (1, 1, 1, 1), mode="reflect"))
    return torch.sigmoid(y) if sigmoid else y


def dense_motion_decoder(S, p, cfg, app, sparse_feats, sparse_motion, sparse_occ, z, training=True):
    """motion_estimator/motion_autoencoder.py:107-149."""
    mp = cfg["model_params"]["motion_estimator"]["dense_motion_decoder"]
    T = cfg["train_params"]["num_predicted_frames"]
    nu = mp["num_up_blocks"]
    n_sparse = cfg["model_params"]["motion_estimator"]["sparse_motion_encoder"]["num_down_blocks"]
    pm = mp["padding_mode"]
    x = block3d(S, p + ".first", z, (1, 1, 1), (1, 1, 1), pm, training)
    fuse_i = 0
    for i in range(nu):
        if i == 0 or not mp["use_appearance_feature"]:
            inp = x
        else:
            a = app[f"enco{nu - i}"]
            a_rep = fold_time(a.unsqueeze(2).repeat(1, 1, T, 1, 1))
            nh, nw = a_rep.shape[-2:]
            if mp["use_feature_resample"]:
                m = resize_flow(fold_time(sparse_motion), [nh, nw])
                o = F.interpolate(fold_time(sparse_occ), size=[nh, nw], mode="bilinear")
                a_rep = resample(a_rep, m) * o
            if list(x.shape[-2:]) != [nh, nw]:
                x = unfold_time(F.interpolate(fold_time(x), size=[nh, nw], mode="bilinear"), x.shape[2])
            inp = torch.cat([x, unfold_time(a_rep, T)], 1)
        x = unfold_time(up_block2d(S, f"{p}.up_blocks.{i}", fold_time(inp), pm, training), 5)
        if i >= nu - n_sparse:
            sf = sparse_feats[f"enco_sparse_{nu - i - 1}"]
            if list(x.shape[-2:]) != list(sf.shape[-2:]):
                x = unfold_time(F.interpolate(fold_time(x), size=list(sf.shape[-2:]), mode="bilinear"), x.shape[2])
            x = block3d(S, f"{p}.fuse_convs.{fuse_i}", torch.cat([x, sf], 1), (1, 1, 1), (1, 1, 1), pm, training)
            fuse_i += 1
    x = unfold_time(up_block2d(S, p + ".final_up_block", fold_time(x), pm, training), 5)
    x = block3d(S, p + ".final_fuse", torch.cat([x, sparse_motion], 1), (1, 1, 1), (1, 1, 1), pm, training)
    x = fold_time(x)
    flow = predictor_head(S, p + ".flow", x, "flow_predictor", False)
    occ = predictor_head(S, p + ".occlusion", x, "occlusion_predictor", True)
    return unfold_time(flow, T), unfold_time(occ, T)


def dense_motion_network(S, cfg, app, mi, rng, training=True, p="motion_encoder"):
    """motion_estimator/dense_motion.py:170-235 (forward; the use_fw_of branch :216-219,226-234 since round 5)."""
    tp, cp = cfg["train_params"], cfg["model_params"]["common_params"]
    t_in, T = tp["num_input_frames"], tp["num_predicted_frames"]
    ap = cfg["model_params"]["appearance_encoder"]
    sf = cp["scale_factor"]
    h_app = int(tp["input_size"][0] / 2 ** ap["num_down_blocks"] * sf)
    w_app = int(tp["input_size"][1] / 2 ** ap["num_down_blocks"] * sf)

    def inputs_then_targets(x):  # :173-192
        src = stack_time_into_channels(x[:, :, :t_in]).unsqueeze(2).repeat(1, 1, T, 1, 1)
        return torch.cat([src, x[:, :, t_in:]], dim=1)

    fr, bg, fg = inputs_then_targets(mi["frames"]), inputs_then_targets(mi["bg_mask"]), inputs_then_targets(mi["fg_mask"])
    inst = inputs_then_targets(mi["instance"])
    flows = torch.cat([mi["target_bw_of"], mi["target_bw_occ"]], dim=1)
    mu_bg, lv_bg = dense_motion_encoder(S, p + ".motion_encoder_bg", cfg, torch.cat([fr, bg, flows], 1).contiguous(), training)
    mu_fg, lv_fg = dense_motion_encoder(S, p + ".motion_encoder_fg", cfg, torch.cat([fr, fg, inst, flows], 1).contiguous(), training)
    out = {"mu": torch.cat([mu_bg, mu_fg], 1), "logvar": torch.cat([lv_bg, lv_fg], 1)}
    z_m = out["mu"] + rng["eps"] * torch.exp(0.5 * out["logvar"])  # :88-92
    thetas = sparse_motion_generator(S, cfg, mi["tracking_gnn"], app["objects_feature"], mi["latent"],
                                     rng["click_index"], training)
    out.update(thetas)
    sparse = generate_sparse_motion(cfg, mi["tracking_gnn"], thetas, mi["instance"][:, :, t_in - 1].float(),
                                    tp["use_gt_training"])
    sparse_feats = sparse_feature_encoder(S, cfg, sparse["sparse_motion_bw"], training)
    code = torch.cat([linear(S, p + ".fc", z_m).view(-1, 64, h_app, w_app), app["app_encoded"]], 1)
    code = same_block_two_conv2d(S, p + ".zconv", code, "reflect")
    codex = app["app_encoded"].unsqueeze(2).repeat(1, 1, T, 1, 1)
    code = torch.cat(torch.chunk(code.unsqueeze(2), T, 1), 2)
    z = torch.cat([codex, code], 1)
    if tp.get("use_fw_of", False):                                                   # :216-219, 226-228: the SAME sparse encoder
        sparse_feats_fw = sparse_feature_encoder(S, cfg, sparse["sparse_motion_fw"], training)
        flow_fw, occ_fw = dense_motion_decoder(S, p + ".dense_generator_fw", cfg, app, sparse_feats_fw,
                                               sparse["sparse_motion_fw"], sparse["sparse_occ_fw"], z, training)
    flow, occ = dense_motion_decoder(S, p + ".dense_generator_bw", cfg, app, sparse_feats,
                                     sparse["sparse_motion_bw"], sparse["sparse_occ_bw"], z, training)
    out.update({k: v for k, v in sparse.items() if not k.startswith("_")})
    out["dense_motion_bw"], out["occlusion_bw"] = flow, occ
    if tp.get("use_fw_of", False):
        out["dense_motion_fw"], out["occlusion_fw"] = flow_fw, occ_fw
    return out


def dense_motion_inference(S, cfg, app, mi, training=False, p="motion_encoder"):
    """motion_estimator/dense_motion.py:236-271 (inference, use_fw_of False): no VAE encoders, z_m and the click index come
    from the caller, the GNN runs in eval mode (model.py:249), thetas are chosen by use_gt_eval."""
    tp, cp = cfg["train_params"], cfg["model_params"]["common_params"]
    t_in, T = tp["num_input_frames"], tp["num_predicted_frames"]
    ap = cfg["model_params"]["appearance_encoder"]
    sf = cp["scale_factor"]
    h_app = int(tp["input_size"][0] / 2 ** ap["num_down_blocks"] * sf)
    w_app = int(tp["input_size"][1] / 2 ** ap["num_down_blocks"] * sf)
    thetas = sparse_motion_generator(S, cfg, mi["tracking_gnn"], app["objects_feature"], mi["latent"],
                                     mi["click_index"], training=False)
    out = dict(thetas)
    sparse = generate_sparse_motion(cfg, mi["tracking_gnn"], thetas, mi["instance"][:, :, t_in - 1].float(),
                                    tp["use_gt_eval"])
    sparse_feats = sparse_feature_encoder(S, cfg, sparse["sparse_motion_bw"], training)
    code = torch.cat([linear(S, p + ".fc", mi["z_m"]).view(-1, 64, h_app, w_app), app["app_encoded"]], 1)
    code = same_block_two_conv2d(S, p + ".zconv", code, "reflect")
    codex = app["app_encoded"].unsqueeze(2).repeat(1, 1, T, 1, 1)
    code = torch.cat(torch.chunk(code.unsqueeze(2), T, 1), 2)
    z = torch.cat([codex, code], 1)
    flow, occ = dense_motion_decoder(S, p + ".dense_generator_bw", cfg, app, sparse_feats,
                                     sparse["sparse_motion_bw"], sparse["sparse_occ_bw"], z, training)
    out.update({k: v for k, v in sparse.items() if not k.startswith("_")})
    out["dense_motion_bw"], out["occlusion_bw"] = flow, occ
    out["index_user_guidance"] = mi["click_index"]
    return out


def flow_embedder(S, p, cfg, x, training=True):
    """generator/flowembedder.py:47-81 (use_decoder True)."""
    fp = cfg["model_params"]["flow_embedder"]
    nd, pm = fp["num_down_blocks"], fp["padding_mode"]
    outs = [same_block2d(S, p + ".conv_first", x, 3, pm, use_norm=False)]
    for i in range(nd):
        outs.append(down_block2d(S, f"{p}.down_blocks.{i}", outs[-1], pm, training))
    if not fp["use_decoder"]:
        return outs
    for i in reversed(range(nd)):
        inp = outs[-1]
        if i != nd - 1:
            tgt = outs[i + 1].shape[-2:]
            if inp.shape[-2:] != tgt:
                inp = F.interpolate(inp, list(tgt), mode="bilinear")
            inp = torch.cat([inp, outs[i + 1]], dim=1)
        outs.append(up_block2d(S, f"{p}.up_blocks.{i}", inp, pm, training))
    return outs[nd:][::-1]


def generator(S, cfg, first_frame, flow, occ, training=True, p="generator", dataset="cityscapes"):
    """generator/generator.py:126-158 (cityscapes branch; both use_spade values)."""
    gp = cfg["model_params"]["generator"]
    nd, pm = gp["num_down_blocks"], gp["padding_mode"]
    if gp["use_spade"]:
        # deform_input (:81-86) mis-reads NCHW as NHWC -> always "resizes" flow to (h, w) of the input: identity here
        fl = F.interpolate(flow, size=first_frame.shape[2:], mode="bilinear")
        img_warp = resample(first_frame, fl)
        feats = flow_embedder(S, p + ".flowembedder", cfg, torch.cat([img_warp, flow, occ], 1), training)
    x = same_block2d(S, p + ".first", first_frame, 7, pm)
    for i in range(nd):
        x = down_block2d(S, f"{p}.down_blocks.{i}", x, pm, training)
    if not gp["use_spade"]:
        fl = F.interpolate(flow, size=x.shape[2:], mode="bilinear")
        warped = resample(x, fl)
        o = occ
        if warped.shape[2:] != o.shape[2:]:
            o = F.interpolate(o, size=warped.shape[2:], mode="bilinear")
        x = warped * o
    for i in range(gp["num_bottleneck_blocks"]):
        x = residual_block(S, f"{p}.middle.{i}", x, training)
    if "kitti" in dataset:                                                          # generator.py:139-145
        xw = same_block2d(S, p + ".first_warped", resample(first_frame, flow), 7, pm)
        for i in range(nd):
            xw = down_block2d(S, f"{p}.down_blocks_warped.{i}", xw, pm, training)
        o = occ
        if xw.shape[2:] != o.shape[2:]:
            o = F.interpolate(o, size=xw.shape[2:], mode="bilinear")
        x = same_block2d(S, p + ".pre_decode.0", torch.cat([x, xw * o], dim=1), 3, pm)
    for i in range(nd):
        if gp["use_spade"]:
            cond = feats[nd - i]
            if x.shape[-2:] != cond.shape[-2:]:
                x = F.interpolate(x, list(cond.shape[-2:]), mode="bilinear")
            x = residual_spade_block(S, f"{p}.up_blocks.{i}", x, cond)
            x = F.interpolate(x, scale_factor=2, mode="bilinear")
        else:
            x = up_block2d(S, f"{p}.up_blocks.{i}", x, pm, training)
    if x.shape[-2:] != first_frame.shape[-2:]:
        x = F.interpolate(x, list(first_frame.shape[-2:]), mode="bilinear")
    return torch.sigmoid(conv2d(S, p + ".final.0", x, 1, 3, "zeros"))


def vgg19_taps(S, p, x, upto="relu5_4"):
    """layers/vgg.py:92-137; returns the relu{1..5}_1 taps (all 16 convs are evaluated, like the reference)."""
    x = (x - S[p + ".mean"]) / S[p + ".std"]
    names = ["relu1_1", "relu1_2", "relu2_1", "relu2_2", "relu3_1", "relu3_2", "relu3_3", "relu3_4",
             "relu4_1", "relu4_2", "relu4_3", "relu4_4", "relu5_1", "relu5_2", "relu5_3", "relu5_4"]
    taps, ni = {}, 0
    for idx, kind, _, _ in thirdparty.vgg19_feature_layout():
        if kind == "pool":
            x = F.max_pool2d(x, 2, 2)
        elif kind == "conv":
            x = F.conv2d(x, S[f"{p}.{names[ni]}.{idx}.weight"], S[f"{p}.{names[ni]}.{idx}.bias"], padding=1)
        else:
            x = F.relu(x)
            taps[names[ni]] = x
            ni += 1
            if ni == len(names):
                break
    return taps


def ssim_loss(x, y):
    """losses/losses.py:152-177."""
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    mu_x, mu_y = F.avg_pool2d(x, 3, 1), F.avg_pool2d(y, 3, 1)
    sx = F.avg_pool2d(x ** 2, 3, 1) - mu_x ** 2
    sy = F.avg_pool2d(y ** 2, 3, 1) - mu_y ** 2
    sxy = F.avg_pool2d(x * y, 3, 1) - mu_x * mu_y
    n = (2 * mu_x * mu_y + c1) * (2 * sxy + c2)
    d = (mu_x ** 2 + mu_y ** 2 + c1) * (sx + sy + c2)
    return torch.clamp((1 - n / d) / 2, 0, 1).mean()


def masked_l1(a, b, mask=None):
    """losses/losses.py:180-189."""
    if mask is not None:
        mask = mask.expand_as(a)
        return F.l1_loss(a * mask, b * mask)
    return F.l1_loss(a, b)


def kl_loss(mu, lv):
    """losses/losses.py:143-149."""
    return (-0.5 * torch.sum(1 + lv - mu.pow(2) - lv.exp())) / mu.numel()


def compute_gram(x):
    """losses/losses.py:32-38."""
    b, ch, h, w = x.size()
    f = x.view(b, ch, w * h)
    return f.bmm(f.transpose(1, 2)) / (h * w * ch)


def perceptual_loss(S, p, gt, fake, T, weights=None):
    """losses/losses.py:23-70: {"perceptual": sum_t sum_l L1(vgg(gt)_l.detach, vgg(fake)_l) / T, "style": the same over the Gram
    matrices of relu2_2 / relu3_4 / relu4_4 / relu5_2} -- each only when its weight is > 0 and the sum is > 0 (:66-69)."""
    weights = weights or {"perceptual": 1, "style": 0}
    content, style = 0.0, 0.0
    for i in range(T):
        a = vgg19_taps(S, p + ".vgg19", gt[:, :, i])
        b = vgg19_taps(S, p + ".vgg19", fake[:, :, i])
        if weights.get("style", 0) > 0:
            for k in ("relu2_2", "relu3_4", "relu4_4", "relu5_2"):
                style = style + F.l1_loss(compute_gram(b[k]), compute_gram(a[k].detach()))
        if weights.get("perceptual", 0) > 0:
            for k in ("relu1_1", "relu2_1", "relu3_1", "relu4_1", "relu5_1"):
                content = content + F.l1_loss(b[k], a[k].detach())
    out = {}
    if torch.is_tensor(content) and float(content.detach()) > 0:
        out["perceptual"] = content / T
    if torch.is_tensor(style) and float(style.detach()) > 0:
        out["style"] = style / T
    return out


def smooth_loss(flow, image):
    """losses/losses.py:73-112 SmoothLoss: edge-aware first differences of each flow component, time folded into the batch.
    (The reference's `gradient_x` differences along H and `gradient_y` along W; kept.)"""
    f, im = fold_time(flow), fold_time(image)

    def gx(t): return t[:, :, :-1, :] - t[:, :, 1:, :]

    def gy(t): return t[:, :, :, :-1] - t[:, :, :, 1:]
    wx = torch.exp(-torch.mean(torch.abs(gx(im)), 1, True))
    wy = torch.exp(-torch.mean(torch.abs(gy(im)), 1, True))
    total = 0
    for c in range(2):
        fc = f[:, c:c + 1]
        total = total + torch.mean(torch.abs(gx(fc) * wx)) + torch.mean(torch.abs(gy(fc) * wy))
    return total / 2


def flow_consist_loss(flow, flowback, mask_fw, mask_bw, T):
    """losses/losses.py:115-140 FlowConsistLoss (masked form; called with the fw / bw occlusion maps at :215-216)."""
    f, fb = fold_time(flow), fold_time(flowback)
    nxt = torch.abs(resample(fb, f) + f)
    prv = torch.abs(resample(f, fb) + fb)
    if mask_bw is not None:
        nxt, prv = fold_time(mask_fw) * nxt, fold_time(mask_bw) * prv
    return (prv.mean() + nxt.mean()) * T


def training_losses(S, cfg, frames, bw_of, bw_occ, gen, gnn, p="objective_func", fw_of=None, fw_occ=None):
    """losses/losses.py:205-255 (the use_fw_of / flow_smooth / style branches since round 5)."""
    tp = cfg["train_params"]
    t_in, T = tp["num_input_frames"], tp["num_predicted_frames"]
    src, tgt = frames[:, :, t_in - 1], frames[:, :, t_in:]
    L = {}
    L["flow_reconstruction"] = masked_l1(gen["dense_motion_bw"], bw_of, bw_occ)
    if fw_of is not None:                                                            # :211-216
        L["flow_reconstruction"] = L["flow_reconstruction"] + masked_l1(gen["dense_motion_fw"], fw_of, fw_occ)
        L["flowcon"] = flow_consist_loss(gen["dense_motion_fw"], gen["dense_motion_bw"], gen["occlusion_fw"],
                                         gen["occlusion_bw"], T)
    warped = torch.stack([resample(src, gen["dense_motion_bw"][:, :, i]) for i in range(T)], 2)
    L["warped"] = masked_l1(warped, tgt)
    if tp["loss_weights"]["flow_smooth"] > 0:                                       # :221-226
        L["flow_smooth"] = smooth_loss(gen["dense_motion_bw"], tgt)
        if fw_of is not None:
            L["flow_smooth"] = L["flow_smooth"] + smooth_loss(gen["dense_motion_fw"], src.unsqueeze(2).repeat(1, 1, T, 1, 1))
    mu, lv = gen["mu"], gen["logvar"]
    L["kl"] = kl_loss(mu, lv)
    L["ssim"] = ssim_loss(fold_time(gen["generated"]), fold_time(tgt))
    L["reconstruction"] = masked_l1(gen["generated"], tgt)
    if tp["loss_weights"]["perceptual"] > 0:
        L.update(perceptual_loss(S, p + ".perceptual_loss", tgt, gen["generated"], T, tp["loss_weights"]))
    L["occlusion_bw"] = masked_l1(bw_occ, gen["occlusion_bw"])
    if fw_of is not None:
        L["occlusion_fw"] = masked_l1(fw_occ, gen["occlusion_fw"])
    tr = sc = ro = 0
    for t in range(T):                                                               # :244-250
        th, gt = gen[f"theta_{t}"], gnn.targets_theta[:, t]
        tr = tr + F.l1_loss(th[:, 2], gt[:, 2]) + F.l1_loss(th[:, 5], gt[:, 5])
        sc = sc + F.l1_loss(th[:, 0], gt[:, 0]) + F.l1_loss(th[:, 4], gt[:, 4])
        ro = ro + F.l1_loss(th[:, 1], gt[:, 1]) + F.l1_loss(th[:, 3], gt[:, 3])
    for name, v in (("translation", tr), ("scale", sc), ("rotation", ro)):
        if torch.any(torch.isnan(v)):                                               # utils/utils.py:375-379
            raise ValueError(f"Value is nan {v}")
        L[name] = v
    return L


def spectral_norm_weight(S, p, training=True):
    """torch.nn.utils.spectral_norm (one power iteration per training forward), as wrapped at discriminator.py:76-77."""
    w = S[p + ".weight_orig"]
    wm = w.reshape(w.shape[0], -1)
    u, v = S[p + ".weight_u"], S[p + ".weight_v"]
    if training:
        with torch.no_grad():
            v = F.normalize(torch.mv(wm.t(), u), dim=0, eps=1e-12)
            u = F.normalize(torch.mv(wm, v), dim=0, eps=1e-12)
            S.t[(p + ".weight_u").lstrip(".")], S.t[(p + ".weight_v").lstrip(".")] = u.clone(), v.clone()
    sigma = torch.dot(u, torch.mv(wm, v))
    return w / sigma


def discriminator(S, p, cfg, x, training=True):
    """discriminator/discriminator.py:59-89 through MultiScaleDiscriminator(num_D=1) :35-56."""
    dp = cfg["model_params"]["discriminator"]
    feats = []
    for i in range(dp["n_layers_D"]):
        x = down_block2d(S, f"{p}.discs.0.down_blocks.{i}", x, dp["padding_mode"], training)
        feats.append(x)
    w = spectral_norm_weight(S, p + ".discs.0.conv", training)
    return feats, F.conv2d(x, w, S[p + ".discs.0.conv.bias"])


def lsgan(pred, real):
    """discriminator.py:96-135 GANLoss: MSE against a constant, on pred[-1] only (last batch element)."""
    last = pred[-1]
    return F.mse_loss(last, torch.full_like(last, 1.0 if real else 0.0))


def d_losses(S, p, cfg, gt, fake, kind, training=True):
    """model.py:101-122 compute_loss_d + gan_and_fm_loss."""
    fr, pr = discriminator(S, p, cfg, gt, training)
    _, pf_det = discriminator(S, p, cfg, fake.detach(), training)
    d_real, d_fake = lsgan(pr, True), lsgan(pf_det, False)
    ff, pf = discriminator(S, p, cfg, fake, training)
    g_gan = lsgan(pf, True)
    fm = 0
    if cfg["train_params"]["loss_weights"][f"feature_matching_{kind}"] > 0:
        for a, b in zip(fr, ff):
            fm = fm + torch.abs(a.detach() - b).mean()
    return d_real, d_fake, g_gan, fm


# --------------------------------------------------------------------------------------------
# the whole step
# --------------------------------------------------------------------------------------------
def forward(S, cfg, batch, rng, training=True):
    """modules/model.py:124-239 GeneratorFullModel.forward (scale_factor 1 => resize_video is a reshape round trip)."""
    tp = cfg["train_params"]
    t_in, T = tp["num_input_frames"], tp["num_predicted_frames"]
    frames, bg, fg = batch["video"], batch["bg_mask"], batch["fg_mask"]
    instance = batch["instance_mask"].float().int()
    input_of, input_occ = batch.get("input_of"), batch.get("input_occ")
    bw_of, bw_occ = batch["target_bw_of"], batch["target_bw_occ"]
    gnn = batch["tracking_gnn"]
    seg = torch.cat([bg[:, :, :t_in], fg[:, :, :t_in]], 1)
    enc_in = torch.cat([stack_time_into_channels(frames[:, :, :t_in]), stack_time_into_channels(seg),
                        stack_time_into_channels(instance[:, :, :t_in])], 1)
    if input_of is not None:
        enc_in = torch.cat([enc_in, stack_time_into_channels(input_of[:, :, :t_in]),
                            stack_time_into_channels(input_occ[:, :, :t_in])], 1)
    app = appearance_encoder(S, cfg, enc_in, gnn, training)
    fw_of, fw_occ = batch.get("target_fw_of"), batch.get("target_fw_occ")           # model.py:148-153 (None unless the data has them)
    mi = dict(frames=frames, bg_mask=bg, fg_mask=fg, instance=instance, target_bw_of=bw_of, target_bw_occ=bw_occ,
              tracking_gnn=gnn, latent=rng["latent_traj"])
    out = dense_motion_network(S, cfg, app, mi, rng, training)
    last = frames[:, :, t_in - 1]
    gen = generator(S, cfg, fold_time(last.unsqueeze(2).repeat(1, 1, T, 1, 1)),
                    fold_time(out["dense_motion_bw"]), fold_time(out["occlusion_bw"]), training)
    out["generated"] = unfold_time(gen, T)
    out["generated_sparse"] = torch.stack(
        [resample(last, out["sparse_motion_bw"][:, :, i].detach()) for i in range(T)], 2).detach()
    out["generated_sparse_occ"] = torch.stack(
        [resample(last, out["sparse_motion_bw"][:, :, i].detach()) * out["sparse_occ_bw"][:, :, i] for i in range(T)], 2)
    loss_g = training_losses(S, cfg, frames, bw_of, bw_occ, out, gnn, fw_of=fw_of, fw_occ=fw_occ)
    loss_d_img, loss_d_vid = {}, {}
    if tp["use_image_discriminator"]:
        dr, df, gg, fm = d_losses(S, "netD_image", cfg, fold_time(batch["video"][:, :, t_in:]),
                                  fold_time(out["generated"]), "image", training)
        loss_g["g_gan_image"], loss_g["feature_matching_image"] = gg, fm
        loss_d_img = {"d_real": dr, "d_fake": df}
    if tp["use_video_discriminator"]:
        fake = torch.cat([stack_time_into_channels(frames[:, :, :t_in]), stack_time_into_channels(out["generated"])], 1)
        dr, df, gg, fm = d_losses(S, "netD_video", cfg, stack_time_into_channels(frames), fake, "video", training)
        loss_g["g_gan_video"], loss_g["feature_matching_video"] = gg, fm
        loss_d_vid = {"d_real": dr, "d_fake": df}
    return out, loss_g, loss_d_img, loss_d_vid


def inference(S, cfg, batch, rng, z_m, training=False):
    """modules/model.py:241-324 GeneratorFullModel.inference: encoder input as in forward, dense_motion.inference, the
    generator on the last input frame, and the two sparse-flow visualisation warps.  `training` is the module mode the
    caller left the model in (trainer.py:201-208 calls it under c2m.eval() + no_grad)."""
    tp = cfg["train_params"]
    t_in, T = tp["num_input_frames"], tp["num_predicted_frames"]
    frames, bg, fg = batch["video"], batch["bg_mask"], batch["fg_mask"]
    instance = batch["instance_mask"].float().int()
    input_of, input_occ = batch.get("input_of"), batch.get("input_occ")
    gnn = batch["tracking_gnn"]
    seg = torch.cat([bg[:, :, :t_in], fg[:, :, :t_in]], 1)
    enc_in = torch.cat([stack_time_into_channels(frames[:, :, :t_in]), stack_time_into_channels(seg),
                        stack_time_into_channels(instance[:, :, :t_in])], 1)
    if input_of is not None:
        enc_in = torch.cat([enc_in, stack_time_into_channels(input_of[:, :, :t_in]),
                            stack_time_into_channels(input_occ[:, :, :t_in])], 1)
    app = appearance_encoder(S, cfg, enc_in, gnn, training)
    mi = dict(instance=instance, latent=rng["latent_traj"], z_m=z_m, click_index=rng["click_index"], tracking_gnn=gnn)
    out = dense_motion_inference(S, cfg, app, mi, training)
    last = frames[:, :, t_in - 1]
    gen = generator(S, cfg, fold_time(last.unsqueeze(2).repeat(1, 1, T, 1, 1)),
                    fold_time(out["dense_motion_bw"]), fold_time(out["occlusion_bw"]), training)
    out["generated"] = unfold_time(gen, T)
    out["generated_sparse"] = torch.stack([resample(last, out["sparse_motion_bw"][:, :, i]) for i in range(T)], 2)
    out["generated_sparse_occ"] = torch.stack(
        [resample(last, out["sparse_motion_bw"][:, :, i]) * out["sparse_occ_bw"][:, :, i] for i in range(T)], 2)
    return out


def train_step_backward(cfg, loss_g, loss_d_img, loss_d_vid):
    """trainer/trainer.py:145-159: weighted generator total, D totals, three backward calls.  Returns the totals."""
    w = cfg["train_params"]["loss_weights"]
    total = torch.tensor(0.0)
    for k, v in loss_g.items():
        total = total + v * w[k]
    tot = {"total_gen": total}
    if loss_d_img:
        tot["total_image_dis"] = (loss_d_img["d_real"] + loss_d_img["d_fake"]) * 0.5
        tot["total_image_dis"].backward()
    if loss_d_vid:
        tot["total_video_dis"] = (loss_d_vid["d_real"] + loss_d_vid["d_fake"]) * 0.5
        tot["total_video_dis"].backward()
    total.backward()
    return tot
