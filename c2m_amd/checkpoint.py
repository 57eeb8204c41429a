"""Checkpoint I/O in the reference's on-disk layout (reference: src/trainer/trainer.py:117-136 load,
:245-260 save): one `torch.save` dict

    {'c2m': state_dict, 'optimizer_gnn': ..., 'optimizer': ..., ['optimizer_d_image': ...], ['optimizer_d_video': ...]}

at `<sampledir>/latest_c2m_model.pth.tar` plus `iter.txt` holding "epoch+1, epoch_iter".  Because c2m_amd keeps the
reference's module tree and torch.optim.Adam's state layout, files written by either side load on the other.
Quirks kept: the model is loaded with strict=False, and 'optimizer_gnn' is saved but NOT restored on resume."""
import os

import numpy as np
import torch


def save_checkpoint(c2m, sampledir, current_epoch, epoch_iter, iter_path=None):
    tp = c2m.train_params
    checkpoint = {"c2m": c2m.state_dict(), "optimizer_gnn": c2m.optimizer_gnn.state_dict(),
                  "optimizer": c2m.optimizer.state_dict()}
    if tp["use_image_discriminator"]:
        checkpoint["optimizer_d_image"] = c2m.d_optimizer_image.state_dict()
    if tp["use_video_discriminator"]:
        checkpoint["optimizer_d_video"] = c2m.d_optimizer_video.state_dict()
    path = os.path.join(sampledir, "latest_c2m_model.pth.tar")
    torch.save(checkpoint, path)
    np.savetxt(iter_path or os.path.join(sampledir, "iter.txt"), (current_epoch + 1, epoch_iter), delimiter=",", fmt="%d")
    return path


def load_checkpoint(c2m, sampledir, which_epoch="latest", local_rank=0, iter_path=None, restore_gnn_optimizer=False):
    """Returns (start_epoch, epoch_iter).  `restore_gnn_optimizer=True` additionally restores the GNN optimizer, which the
    reference forgets to do (trainer.py:124-129)."""
    tp = c2m.train_params
    path = os.path.join(sampledir, f"{which_epoch}_c2m_model.pth.tar")
    map_location = {"cuda:0": f"cuda:{local_rank}"} if torch.cuda.is_available() else "cpu"
    state = torch.load(path, map_location=map_location, weights_only=False)
    c2m.load_state_dict(state["c2m"], strict=False)
    c2m.optimizer.load_state_dict(state["optimizer"])
    if restore_gnn_optimizer and "optimizer_gnn" in state:
        c2m.optimizer_gnn.load_state_dict(state["optimizer_gnn"])
    if tp["use_image_discriminator"]:
        c2m.d_optimizer_image.load_state_dict(state["optimizer_d_image"])
    if tp["use_video_discriminator"]:
        c2m.d_optimizer_video.load_state_dict(state["optimizer_d_video"])
    try:
        start_epoch, epoch_iter = np.loadtxt(iter_path or os.path.join(sampledir, "iter.txt"), delimiter=",", dtype=int)
    except FileNotFoundError:
        start_epoch, epoch_iter = 1, 0
    return int(start_epoch), int(epoch_iter)
