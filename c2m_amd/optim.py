"""Adam on the HIP multi-tensor kernel (reference: src/modules/model.py:54-99 builds four
torch.optim.Adam(betas=(0.5, 0.999), eps=1e-7) + MultiStepLR, src/trainer/trainer.py:155-165 steps them).

`Adam` subclasses torch.optim.Adam and keeps its state layout ('step' / 'exp_avg' / 'exp_avg_sq' per parameter), so
state_dict()/load_state_dict() exchange checkpoints with the reference and torch LR schedulers drive `param_groups`
unchanged; only `step()` is replaced: ONE kernel launch per parameter group instead of ~8 elementwise launches per group
per foreach op.  No CPU fallback: parameters must live on a HIP device."""
import math

import numpy as np
import torch

from . import _lib
from .ops import _p, _stream


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise ValueError("c2m_amd.optim.Adam implements the reference's configuration: no weight decay, no amsgrad")
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, foreach=False,
                         fused=False, capturable=False, differentiable=False, maximize=False)
        self._tables = {}
        self._plans = {}

    def _table(self, tensors):
        """Device pointer table + block map for one set of (param, grad, exp_avg, exp_avg_sq); cached on the pointers."""
        key = tuple(t.data_ptr() for quad in tensors for t in quad)
        hit = self._tables.get(key)
        if hit is not None:
            return hit
        if len(self._tables) > 16:
            self._tables.clear()
        chunk = _lib.lib().c2m_adam_chunk()
        n = len(tensors)
        ptrs = np.empty((4, n), dtype=np.int64)
        sizes = np.empty(n, dtype=np.int64)
        bm = []
        for i, quad in enumerate(tensors):
            for j, t in enumerate(quad):
                ptrs[j, i] = t.data_ptr()
            sizes[i] = quad[0].numel()
            nch = (quad[0].numel() + chunk - 1) // chunk
            bm.append(np.stack([np.full(nch, i, dtype=np.int32), np.arange(nch, dtype=np.int32)], axis=1))
        bm = np.concatenate(bm, axis=0)
        dev = tensors[0][0].device
        hit = (torch.from_numpy(ptrs).to(dev), torch.from_numpy(sizes).to(dev), torch.from_numpy(bm).to(dev), n, len(bm))
        self._tables[key] = hit
        return hit

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        for gi, group in enumerate(self.param_groups):
            beta1, beta2 = group["betas"]
            plist = [p for p in group["params"] if p.grad is not None]
            if not plist:
                continue
            # fast path: same tensors as the previous step (pointers unchanged) -> reuse the validated launch plan
            sig = tuple(p.grad.data_ptr() for p in plist) + tuple(p.data_ptr() for p in plist)
            plan = self._plans.get(gi)
            if plan is None or plan["sig"] != sig or any(self.state[p].get("exp_avg") is not e
                                                         for p, e in zip(plist, plan["exp_avg"])):
                plan = self._plans[gi] = self._build_plan(plist, sig)
            torch._foreach_add_(plan["steps"], 1)
            for sub in plan["subs"]:
                sub["step"] += 1
                step = sub["step"]
                bc1 = 1.0 - beta1 ** step
                bc2 = 1.0 - beta2 ** step
                table, sizes, bm, n, nblocks = sub["table"]
                _lib.check(L.c2m_adam_step(_p(table), _p(sizes), _p(bm), n, nblocks, beta1, beta2, group["eps"],
                                           group["lr"] / bc1, math.sqrt(bc2), _stream()), "adam_step")
            # the kernel writes through raw pointers: tell autograd (saved-tensor checks) and every `_version`-keyed
            # cache (ops._packed: packed / Winograd-transformed copies of frozen weights) that the values changed
            torch.autograd.graph.increment_version(plist)
        # every cached pack (K-order matrix, bf16 patch image) of the weights just updated: rebuilt in place by ONE launch instead of
        # one launch per layer and layout at their next use
        from . import ops
        ops.refresh_trainable_packs([p for group in self.param_groups for p in group["params"]])
        return loss

    def _build_plan(self, plist, sig):
        by_step = {}
        steps, exp_avgs = [], []
        for p in plist:
            if not p.is_cuda or p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                raise RuntimeError("c2m_amd.optim.Adam: fp32 parameters on a HIP device only (no CPU fallback)")
            if p.grad.is_sparse:
                raise RuntimeError("Adam does not support sparse gradients")
            if not (p.is_contiguous() and p.grad.is_contiguous()):
                raise RuntimeError("c2m_amd.optim.Adam: parameters and gradients must be contiguous")
            st = self.state[p]
            if len(st) == 0:                     # same lazy state as torch.optim.Adam._init_group
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if st["step"].is_cuda:               # a checkpoint saved by a capturable/fused torch Adam
                st["step"] = st["step"].cpu()
            steps.append(st["step"])
            exp_avgs.append(st["exp_avg"])
            by_step.setdefault(float(st["step"]), []).append((p, p.grad, st["exp_avg"], st["exp_avg_sq"]))
        subs = [dict(step=s, table=self._table(t)) for s, t in by_step.items()]
        return dict(sig=sig, steps=steps, exp_avg=exp_avgs, subs=subs)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plans.clear()
