"""Data-parallel gradient synchronisation over RCCL/xGMI (torch.distributed backend "nccl" IS RCCL on ROCm).

The reference wraps the model in DistributedDataParallel and immediately unwraps it (src/train.py:80-85), so its
gradients are never synchronised.  This reducer implements the intended semantics -- every rank ends
each step with the mean-of-ranks gradient -- for a model that is called *directly* (three backward calls per step,
four optimizers, ~2.3 M parameters that never receive a gradient):

  * parameters are packed into flat fp32 buckets in reverse registration order; `.grad` of every parameter is a view
    into its bucket, so autograd accumulates straight into the communication buffer (no copy-in / copy-out);
  * `arm()` is called before the LAST backward of the step; from then on a bucket is all-reduced on a side HIP stream
    as soon as all of its parameters that are expected to fire have fired, overlapping RCCL with the rest of backward;
    the expected set is learned from the first step (which is reduced at `finish()` without overlap);
  * `finish()` flushes what is left, makes the compute stream wait for the side stream and sets `.grad = None` for
    parameters that received no gradient this step (the optimizers then skip them exactly as in the reference).
On CPU tensors (gloo, used by the tests) the same logic runs synchronously.
"""
import torch
import torch.distributed as dist


class _Bucket:
    def __init__(self, params, device):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.flat = torch.zeros(self.numel, device=device, dtype=torch.float32)
        self.views, off = [], 0
        for p in params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.work = None
        self.launched = False
        self.pending = 0


class GradientReducer:
    def __init__(self, params, bucket_mb=25.0, process_group=None, force_collectives=False):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force_collectives: issue the all-reduces even in a 1-rank group (exercises the RCCL / side-stream path on a
        # single GPU; used by the tests)
        self.collectives = self.world > 1 or (force_collectives and dist.is_initialized())
        params = [p for p in params if p.requires_grad]
        seen, uniq = set(), []
        for p in params:
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        self.params = uniq
        if not uniq:
            raise ValueError("no trainable parameters")
        self.device = uniq[0].device
        self.on_gpu = self.device.type == "cuda"
        cap = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets, cur, n = [], [], 0
        for p in reversed(uniq):                      # gradients become ready roughly in reverse creation order
            if cur and n + p.numel() > cap:
                self.buckets.append(_Bucket(cur, self.device))
                cur, n = [], 0
            cur.append(p)
            n += p.numel()
        if cur:
            self.buckets.append(_Bucket(cur, self.device))
        self.where = {}
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self.where[id(p)] = (b, i)
        self.stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self.expected = None          # ids of params that fire in the final backward (learned on step 1)
        self.fired_final, self.fired_any = set(), set()
        self.armed = False
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in uniq]
        self.zero_grad()

    # ---- step protocol -------------------------------------------------------------------------------------
    def zero_grad(self):
        """Replaces optimizer.zero_grad(): zero the buckets and (re)bind every .grad to its bucket view."""
        for b in self.buckets:
            b.flat.zero_()
            b.work, b.launched = None, False
            for p, v in zip(b.params, b.views):
                p.grad = v
        self.fired_final.clear()
        self.fired_any.clear()
        self.armed = False

    def arm(self):
        """Call right before the last backward() of the step."""
        self.armed = True
        self.fired_final.clear()
        if self.expected is not None:
            for b in self.buckets:
                b.pending = sum(1 for p in b.params if id(p) in self.expected)

    def _on_grad(self, p):
        b, i = self.where[id(p)]
        if p.grad is not b.views[i] and p.grad.data_ptr() != b.views[i].data_ptr():
            b.views[i].copy_(p.grad)              # someone reset .grad (e.g. zero_grad(set_to_none=True)): re-adopt
            p.grad = b.views[i]
        self.fired_any.add(id(p))
        if not self.armed:
            return
        self.fired_final.add(id(p))
        if self.expected is not None and id(p) in self.expected and not b.launched:
            b.pending -= 1
            if b.pending == 0:
                self._launch(b)

    def _launch(self, b):
        b.launched = True
        if not self.collectives:
            return
        if self.on_gpu:
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.stream):
                b.flat.div_(self.world)
                b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            b.flat.div_(self.world)
            dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group)

    def finish(self):
        """After the last backward: reduce the remaining buckets, join streams, drop never-touched grads."""
        for b in self.buckets:
            if not b.launched:
                self._launch(b)
        if self.on_gpu and self.collectives:
            for b in self.buckets:
                if b.work is not None:
                    b.work.wait()
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
        if self.expected is None:
            self.expected = set(self.fired_final)
        for p in self.params:
            if id(p) not in self.fired_any:
                p.grad = None

    def bytes_per_step(self):
        return 4 * sum(b.numel for b in self.buckets)

    def remove(self):
        for h in self._hooks:
            h.remove()
