"""Config surface of the hot path.

The reference reads its YAML straight into a dict (src/train.py:50-51) and indexes it with plain
keys everywhere, so the "API" is the key set of src/config/c2m_journal_cityscapes.yaml:35-157
(train_params / model_params).  `load_config` accepts any YAML with that surface (the reference's
own file drops in unchanged); `default_config` builds the same dict programmatically so that
bench/tests need no file.  One deviation, recorded in SURVEY.md App. A.1: the shipped YAML lacks
`model_params.generator.use_spade` (generator.py:21 -> KeyError); we default it to True.
"""
import copy

import yaml


def default_config(height=128, width=256, num_input_frames=1, block_expansion=32, max_expansion=512,
                   h_dim=512, z_dim=1024, out_channel=512, ndf=32, use_image_discriminator=True,
                   use_video_discriminator=True, use_spade=True, batch_size=3):
    """Same keys/values as c2m_journal_cityscapes.yaml:35-157 (width knobs exposed for small test nets)."""
    be, me = block_expansion, max_expansion
    train_params = dict(
        num_input_frames=num_input_frames, num_predicted_frames=5, input_size=[height, width], num_epochs=300,
        lr_rate_d=4.0e-4, lr_rate_g=2.0e-4, lr_rate_gnn=1.0e-4, gamma_d=0.8, gamma_g=0.9, gamma_gnn=1.0,
        milestone_start=100, milestone_end=400, milestone_every=50, seed=31415, batch_size=batch_size, workers=4,
        local_world_size=2, use_gt_training=True, use_gt_eval=False, use_pre_processed_of=True, use_fw_of=False,
        beta1=0.5, beta2=0.999, eps=1e-7, continue_train=False,
        use_image_discriminator=use_image_discriminator, use_video_discriminator=use_video_discriminator,
        eval_freq=4600,
        loss_weights=dict(flow_reconstruction=10, flow_smooth=0, flowcon=0, reconstruction=100, kl=100, ssim=10,
                          perceptual=10, occlusion_bw=20, occlusion_fw=20, g_gan_image=1, g_gan_video=1,
                          feature_matching_image=10, feature_matching_video=10, warped=100, scale=2, rotation=1,
                          translation=100))
    model_params = dict(
        common_params=dict(scale_factor=1, image_channel=3, seg_channel_bg=11, seg_channel_fg=9,
                           instance_channel=1, flow_channel=2, occlusion_channel=1),
        motion_estimator=dict(
            sparse_motion_estimator=dict(h_dim=h_dim, z_dim=z_dim, num_features_x=23, num_features_y=6),
            sparse_motion_encoder=dict(block_expansion=be, num_down_blocks=4, max_expansion=me, in_channel=2,
                                       padding_mode="reflect"),
            dense_motion_encoder=dict(out_channel_bg=out_channel, out_channel_fg=out_channel, max_expansion=me,
                                      block_expansion=be, num_down_blocks=6, padding_mode="reflect",
                                      t_kernel=[4, 3, 3, 4, 1, 1, 1], h_kernel=[4, 4, 4, 4, 4, 4, 3],
                                      w_kernel=[4, 4, 4, 4, 4, 4, 3], t_stride=[2, 1, 1, 2, 1, 1, 1],
                                      h_stride=[2, 2, 2, 2, 2, 2, 1], w_stride=[2, 2, 2, 2, 2, 2, 1],
                                      t_padding=[1, 1, 1, 1, 0, 0, 0], h_padding=[1, 1, 1, 1, 1, 1, 1],
                                      w_padding=[1, 1, 1, 1, 1, 1, 1]),
            dense_motion_decoder=dict(in_channel=min(me, be * 32) + 16, out_channel=be, block_expansion=be,
                                      max_expansion=me, num_up_blocks=5, padding_mode="reflect",
                                      use_appearance_feature=True, use_feature_resample=True)),
        discriminator=dict(in_channel=3, ndf=ndf, n_layers_D=4, num_D=1, padding_mode="reflect"),
        appearance_encoder=dict(block_expansion=be, num_down_blocks=6, max_expansion=me, pooling_after=2,
                                padding_mode="reflect", pool_size=7),
        generator=dict(block_expansion=be, num_down_blocks=3, max_expansion=me, num_bottleneck_blocks=4,
                       padding_mode="reflect", use_skip=False, use_spade=use_spade),
        flow_embedder=dict(input_channel=6, block_expansion=be, num_down_blocks=3, max_expansion=me,
                           padding_mode="reflect", use_decoder=True))
    return dict(name="c2m_journal", suffix="", dataset_params=dict(dataset="cityscapes"),
                train_params=train_params, model_params=model_params)


def normalize_config(cfg):
    """Fill the one key the shipped YAML forgets; returns a deep copy (ctors mutate their dicts)."""
    cfg = copy.deepcopy(cfg)
    cfg["model_params"]["generator"].setdefault("use_spade", True)
    cfg["train_params"]["eps"] = float(cfg["train_params"]["eps"])
    return cfg


def load_config(path):
    with open(path) as f:
        return normalize_config(yaml.safe_load(f))
