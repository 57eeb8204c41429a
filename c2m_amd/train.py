"""Training-step harness: our counterpart of the reference's Trainer.update_model (src/trainer/trainer.py:138-168)
and of the process setup in src/train.py:141-159.  Same loss weighting, same three backward calls in the same order
(D_image, D_video, generator total -- the generator backward also deposits gradients in D parameters, reproduced),
same four Adam steps; plus the real mean-of-ranks gradient all-reduce the reference lacks (c2m_amd/ddp.py)."""
import os

import torch
import torch.distributed as dist

from . import ops
from .ddp import GradientReducer


def init_distributed():
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); backend nccl == RCCL on ROCm."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a one-GPU box: every rank on device 0, collectives over gloo (RCCL refuses two ranks on one device);
    # exercises the multi-rank reducer logic on HIP tensors -- never used by the real launch
    shared = bool(os.environ.get("C2M_REHEARSAL_SHARED_GPU"))
    if shared:
        local_rank = 0
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    if (world > 1 or os.environ.get("C2M_FORCE_PROCESS_GROUP")) and "MASTER_ADDR" in os.environ \
            and not dist.is_initialized():
        dist.init_process_group(backend="nccl" if torch.cuda.is_available() and not shared else "gloo")
    return int(os.environ.get("RANK", "0")), local_rank, world


def compute_flow(flownet, train_batch, train_params):
    """Trainer.compute_flow (src/trainer/trainer.py:42-98): optical flow + occlusion targets from the frozen flow net when
    `use_pre_processed_of` is False.  Forward flow between consecutive input frames -> `input_of` / `input_occ` (the
    occlusion comes from the reverse pair); between the last input frame and every predicted frame both directions ->
    `target_bw_of` / `target_bw_occ` (and `target_fw_*` when `use_fw_of`).  Frames are mapped to [-1, 1] first.
    The 2 * (t_in - 1 + t_out) flow-net passes of the reference loop are issued as ONE batched pass: the pairs are
    independent images and every kernel of the net is batch-wise bit-deterministic."""
    t_in, t_out = train_params["num_input_frames"], train_params["num_predicted_frames"]
    video = train_batch["video"]
    b = video.shape[0]
    frame = lambda i: video[:, :, i] * 2 - 1
    pairs = []                                           # (a, b) frame indices: flow a -> b, occlusion map of that flow
    for i in range(t_in - 1):
        pairs += [(i, i + 1), (i + 1, i)]
    for i in range(t_out):
        pairs += [(t_in - 1, t_in + i), (t_in + i, t_in - 1)]
    A = torch.cat([frame(a) for a, _ in pairs], 0)
    B = torch.cat([frame(c) for _, c in pairs], 0)
    flow, conf = flownet(A, B)
    flow = [f.unsqueeze(2) for f in flow.split(b)]
    conf = [c.unsqueeze(2) for c in conf.split(b)]
    n_in = t_in - 1
    out = {}
    # input frames: forward flow of (i -> i+1), confidence of the REVERSE pair (trainer.py:63-69)
    out["input_of"] = torch.cat([flow[2 * i] for i in range(n_in)], dim=2) if n_in else None
    out["input_occ"] = torch.cat([conf[2 * i + 1] for i in range(n_in)], dim=2) if n_in else None
    fw = [2 * n_in + 2 * i for i in range(t_out)]      # (last input -> target i); +1: the reverse pair
    out["target_bw_of"] = torch.cat([flow[k + 1] for k in fw], dim=2)
    out["target_bw_occ"] = torch.cat([conf[k] for k in fw], dim=2)
    if train_params.get("use_fw_of", False):
        out["target_fw_of"] = torch.cat([flow[k] for k in fw], dim=2)
        out["target_fw_occ"] = torch.cat([conf[k + 1] for k in fw], dim=2)
    return out


class TrainStep:
    def __init__(self, c2m, loss_weights=None, run_optimizers=True, distributed=None, bucket_mb=25.0,
                 force_collectives=False, measure_comm=False, comm_dtype=torch.float32, measure_comm_buckets=False):
        self.c2m = c2m
        self.tp = c2m.train_params
        self.loss_weights = loss_weights or self.tp["loss_weights"]
        self.run_optimizers = run_optimizers
        self.optimizers = [c2m.optimizer, c2m.optimizer_gnn]
        if self.tp["use_image_discriminator"]:
            self.optimizers.append(c2m.d_optimizer_image)
        if self.tp["use_video_discriminator"]:
            self.optimizers.append(c2m.d_optimizer_video)
        distributed = dist.is_initialized() and dist.get_world_size() > 1 if distributed is None else distributed
        # leaf weights / biases of the convolution modules: their gradients are deferred to the side stream (ops.deferred_wgrads),
        # the reducer adopts them instead of summing them into its buckets on the backward's stream
        conv_params = [p for m in c2m.modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.Conv3d))
                       for p in (m._parameters.get("weight"), m._parameters.get("bias")) if p is not None and p.requires_grad]
        self.reducer = GradientReducer(list(c2m.parameters()), bucket_mb=bucket_mb, buffers=list(c2m.buffers()),
                                       force_collectives=force_collectives, measure=measure_comm, measure_buckets=measure_comm_buckets,
                                       comm_dtype=comm_dtype, adopt_params=conv_params) if distributed else None
        self._deferred_nan = []       # NaN checks recorded while capturing (utils.isnan), evaluated after every replay

    # ---- HIP-graph replay of zero_grad + forward + backward (single GPU, static batch) ---------------------------------------
    def capture(self, data, warmup=3):
        """Capture one update WITHOUT the optimizer steps into a HIP graph for the static batch `data`; afterwards
        `__call__(data)` with the same object replays the graph (one launch instead of ~1200) and then runs the optimizers
        eagerly (their bias-correction scalars are host-computed kernel arguments and change every step).

        New data for a captured step must be copied INTO the tensors of `data` (static addresses).  With the gradient
        reducer (N > 1) the graph holds zero_grad + forward + backward only -- gradients accumulate into the reducer's flat
        buckets (static addresses) -- and the RCCL all-reduces are issued eagerly after each replay (`reduce_all`, no
        overlap with backward: hooks do not run in a replay); every rank must call capture() at the same point.
        Everything on the captured path is kernel nodes only: the one hipMemsetAsync per splat / warp-inversion of round 1
        was a memset NODE whose replay faulted ("write access to a read-only page" on the second replay, ROCm 7.2) and is a
        zero-fill kernel now; the reference's NaN checks (host syncs, losses.py:251-253) are recorded during the capture
        and evaluated after every replay (`check_nan_every`: one fused reduction, synchronised every that many steps)."""
        dev = next(self.c2m.parameters()).device
        # An AccumulateGrad node lives on the stream it was created on and stays alive while ANY autograd graph refers to it
        # (nn.utils.spectral_norm, for one, keeps `module.weight = weight_orig / sigma` -- a tensor with grad_fn -- from one
        # forward to the next).  After eager steps on ANOTHER stream the capture would need cross-stream syncs, which are
        # illegal while capturing (segfault in capture_end on ROCm 7.2).  So: run every eager step that precedes a capture
        # under `with torch.cuda.stream(step.graph_stream)` (bench.py --graph does), or capture before the first eager step.
        side = self.graph_stream
        side.wait_stream(torch.cuda.current_stream(dev))
        from .utils import utils as U
        run_opt, self.run_optimizers = self.run_optimizers, False
        try:
            with torch.cuda.stream(side):
                for _ in range(warmup):                       # plans, caches and allocator state settle outside the capture
                    self._eager(data)                         # (with a reducer: also learns the rank-agreed gradient sets)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            U.begin_deferred_nan()
            from . import thirdparty
            thirdparty.reset_capture_cache()                  # tensors cached during an earlier capture belong to that graph's pool
            try:
                # With a process group alive its watchdog THREAD polls the completion events of earlier collectives; in the default
                # ("global") capture mode any such HIP call from another thread during the capture is an error that takes the
                # process down ("operation not permitted when stream is capturing" -- seen once in ~6 runs of
                # test_bench_graph_replay_with_the_reducer).  Thread-local mode restricts the check to the capturing thread.
                mode = "thread_local" if dist.is_available() and dist.is_initialized() else "global"
                with torch.cuda.graph(graph, stream=side, capture_error_mode=mode):
                    outputs = self._eager(data, in_capture=True)
            finally:
                self._deferred_nan = U.end_deferred_nan()
                thirdparty.reset_capture_cache()
        finally:
            self.run_optimizers = run_opt
        self._graph, self._graph_data, self._graph_out = graph, data, outputs
        self._replays = 0
        return self

    check_nan_every = 1               # replays between two host-side evaluations of the deferred NaN checks

    def _check_deferred_nan(self):
        """The reference raises ValueError when a theta loss is NaN (utils.py:375-379 <- losses.py:251-253).  Under replay
        the flags live in static graph tensors: one fused any() over them, read back every `check_nan_every` replays."""
        if not self._deferred_nan:
            return
        self._replays += 1
        if self._replays % max(1, int(self.check_nan_every)):
            return
        flags = torch.stack([f.reshape(()) for f, _ in self._deferred_nan])
        if bool(flags.any()):
            bad = [name for (f, name) in self._deferred_nan if bool(f)]
            raise ValueError(f"deferred check failed after a HIP-graph replay (nan / captured-layout guard) in {', '.join(bad)}")

    @property
    def graph_stream(self):
        """The (non-default) stream captures are recorded on; eager steps that precede a capture must run on it too."""
        if getattr(self, "_graph_stream", None) is None:
            self._graph_stream = torch.cuda.Stream(device=next(self.c2m.parameters()).device)
        return self._graph_stream

    def zero_grad(self):
        if self.reducer is not None:
            self.reducer.zero_grad()
        else:
            for o in self.optimizers:
                o.zero_grad(set_to_none=True)

    def __call__(self, data):
        """One update; returns (generated dict, generator loss dict incl. total_gen, D loss dict)."""
        if getattr(self, "_graph", None) is not None and data is self._graph_data:
            ops.refresh_trainable_packs()       # weights changed outside Adam.step (load_state_dict ...): the graph holds no pack kernels
            self._graph.replay()
            if self.reducer is not None:
                self.reducer.reduce_all()
            self._check_deferred_nan()
            if self.run_optimizers:
                for o in self.optimizers:
                    o.step()
            return self._graph_out
        return self._eager(data)

    def _eager(self, data, in_capture=False):
        self.zero_grad()
        generated, loss_g, loss_d_img, loss_d_vid = self.c2m(data)
        total = None
        for k, v in loss_g.items():
            term = v * self.loss_weights[k]
            total = term if total is None else total + term
        loss_g["total_gen"] = total
        losses_d = {}
        if self.tp["use_image_discriminator"]:
            losses_d["total_image_dis"] = (loss_d_img.get("d_real", 0) + loss_d_img.get("d_fake", 0)) * 0.5
            losses_d["total_image_dis"].backward()
        if self.tp["use_video_discriminator"]:
            losses_d["total_video_dis"] = (loss_d_vid.get("d_real", 0) + loss_d_vid.get("d_fake", 0)) * 0.5
            losses_d["total_video_dis"].backward()
        if self.reducer is not None and not in_capture:
            self.reducer.arm()
        # weight gradients on the side stream, joined once after the backward (ops.deferred_wgrads); with the gradient reducer the
        # hook moves an adopted gradient into its bucket on that stream, and a bucket's all-reduce waits for every branch stream
        with ops.deferred_wgrads():
            total.backward()
        if self.reducer is not None and not in_capture:
            self.reducer.finish()
        if self.run_optimizers:
            for o in self.optimizers:
                o.step()
        return generated, loss_g, losses_d
