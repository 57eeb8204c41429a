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
        self.ready = False
        self.pending = 0


class GradientReducer:
    def __init__(self, params, bucket_mb=25.0, process_group=None, force_collectives=False, buffers=(),
                 measure=False):
        """buffers: module buffers (BatchNorm running statistics, spectral-norm u/v) that are broadcast from rank 0
        together with the parameters at construction (what DistributedDataParallel's constructor does); afterwards they
        stay per-rank like the reference's BatchNorm.  measure: record HIP events around the join in finish() so that
        `exposed_ms()` reports the all-reduce time that was NOT hidden behind backward."""
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
        self.expected = None          # ids of params that fire in the final backward (learned on step 1, rank-agreed)
        self.ever_fired = None        # ids of params that receive a gradient at all (learned on step 1, rank-agreed)
        self.fired_final, self.fired_any = set(), set()
        self.armed = False
        self.next_bucket = 0          # buckets are all-reduced strictly in index order: same RCCL call sequence on all ranks
        self.measure = measure and self.on_gpu
        self._exposed = []            # (event at end of backward compute, event after the side stream was joined)
        if self.world > 1:            # identical starting point on every rank, whatever the callers seeded
            with torch.no_grad():
                for t in list(uniq) + [b for b in buffers if torch.is_tensor(b)]:
                    dist.broadcast(t.data, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                                   group=self.group)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in uniq]
        self.zero_grad()

    # ---- step protocol -------------------------------------------------------------------------------------
    def zero_grad(self):
        """Replaces optimizer.zero_grad(): zero the buckets and (re)bind every .grad to its bucket view."""
        for b in self.buckets:
            b.flat.zero_()
            b.work, b.launched, b.ready = None, False, False
            for p, v in zip(b.params, b.views):
                p.grad = v
        self.fired_final.clear()
        self.fired_any.clear()
        self.armed = False
        self.next_bucket = 0

    def arm(self):
        """Call right before the last backward() of the step."""
        self.armed = True
        self.fired_final.clear()
        if self.expected is not None:
            for b in self.buckets:
                b.pending = sum(1 for p in b.params if id(p) in self.expected)
                b.ready = b.pending == 0
            self._launch_ready()

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
                b.ready = True
                self._launch_ready()

    def _launch_ready(self):
        while self.next_bucket < len(self.buckets) and self.buckets[self.next_bucket].ready:
            self._launch(self.buckets[self.next_bucket])
            self.next_bucket += 1

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
        ev0 = None
        if self.measure and self.collectives:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record(torch.cuda.current_stream(self.device))
        for b in self.buckets:                         # index order (see next_bucket)
            if not b.launched:
                self._launch(b)
        self.next_bucket = len(self.buckets)
        if self.on_gpu and self.collectives:
            for b in self.buckets:
                if b.work is not None:
                    b.work.wait()
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
            if ev0 is not None:
                ev1 = torch.cuda.Event(enable_timing=True)
                ev1.record(torch.cuda.current_stream(self.device))
                self._exposed.append((ev0, ev1))
        if self.expected is None:
            # Step 1: agree across ranks on which parameters fire (in the final backward / at all).  Every later step
            # uses the agreed sets, so all ranks issue the same bucket sequence and drop the same gradients even if a
            # data-dependent branch made one rank's graph differ.
            fin = torch.tensor([[float(id(p) in self.fired_final), float(id(p) in self.fired_any)] for p in self.params],
                               device=self.device)
            if self.world > 1:
                dist.all_reduce(fin, op=dist.ReduceOp.MAX, group=self.group)
            fin = fin.cpu()
            self.expected = {id(p) for p, f in zip(self.params, fin) if f[0] > 0}
            self.ever_fired = {id(p) for p, f in zip(self.params, fin) if f[1] > 0}
        for p in self.params:
            if id(p) not in self.fired_any and id(p) not in self.ever_fired:
                p.grad = None

    def exposed_ms(self):
        """Mean GPU time per step between the end of the backward kernels and the moment the compute stream may go on
        (all-reduce time that the overlap did not hide); None unless constructed with measure=True on a GPU."""
        if not self._exposed:
            return None
        torch.cuda.synchronize(self.device)
        return sum(a.elapsed_time(b) for a, b in self._exposed) / len(self._exposed)

    def reset_measurements(self):
        self._exposed = []

    def bytes_per_step(self):
        return 4 * sum(b.numel for b in self.buckets)

    def remove(self):
        for h in self._hooks:
            h.remove()
