"""Data-parallel gradient synchronisation over RCCL/xGMI (torch.distributed backend "nccl" IS RCCL on ROCm).

The reference wraps the model in DistributedDataParallel and immediately unwraps it (src/train.py:80-85), so its
gradients are never synchronised.  This reducer implements the intended semantics -- every rank ends
each step with the mean-of-ranks gradient -- for a model that is called *directly* (three backward calls per step,
four optimizers, ~2.3 M parameters that never receive a gradient):

  * parameters are packed into flat fp32 buckets in reverse registration order; `.grad` of every parameter is a view
    into its bucket, so autograd accumulates straight into the communication buffer (no copy-in / copy-out);
  * `arm()` is called before the LAST backward of the step; from then on a bucket is all-reduced on a side HIP stream
    as soon as all of its parameters that are expected to fire have fired, overlapping RCCL with the rest of backward.
    Buckets are launched in a RANK-AGREED ORDER, the order in which they completed on rank 0 during the first step
    (not bucket-index order: one late parameter in bucket k would otherwise hold back every later bucket); every rank
    issues the same RCCL call sequence, whatever its own completion order was.  The expected set and the order are
    learned on the first step (which is reduced at `finish()` without overlap);
  * the collective is `ReduceOp.AVG` on RCCL (one pass; gloo has no AVG: pre-divide + SUM there).  `comm_dtype=
    torch.bfloat16` (BASELINE configs[3-4], SURVEY 8d: 212 MB instead of 424 MB per step) sends a bf16 copy of each
    bucket and expands the result back into the fp32 bucket -- the accumulation across the three backward calls and
    the optimizer input stay fp32;
  * `finish()` flushes what is left, makes the compute stream wait for the side stream and sets `.grad = None` for
    parameters that received no gradient this step (the optimizers then skip them exactly as in the reference);
  * `reduce_all()` is the no-overlap form for a HIP-graph replay of forward + backward (hooks do not run in a replay):
    every bucket is reduced after the replay, eagerly, in bucket order.
On CPU tensors (gloo, used by the tests) the same logic runs synchronously.
"""
import torch
import torch.distributed as dist

from . import ops


class _Bucket:
    def __init__(self, params, device, comm_dtype):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.flat = torch.zeros(self.numel, device=device, dtype=torch.float32)
        self.comm = None if comm_dtype == torch.float32 else torch.empty(self.numel, device=device, dtype=comm_dtype)
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
                 measure=False, comm_dtype=torch.float32, measure_buckets=False, adopt_params=()):
        """params: ALL parameters of the model.  The trainable ones are bucketed; every parameter -- frozen ones (the
        VGG-19 of the perceptual loss, a flow net) included -- and every tensor of `buffers` (BatchNorm running
        statistics, spectral-norm u/v) is broadcast from rank 0 at construction, which is what DistributedDataParallel's
        `_sync_module_states` does: with per-rank seeding the ranks would otherwise optimise different perceptual losses
        while averaging their gradients.  Afterwards buffers stay per-rank like the reference's BatchNorm.
        measure: record HIP events around the join in finish() so that `exposed_ms()` reports the all-reduce time that
        was NOT hidden behind backward."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force_collectives: issue the all-reduces even in a 1-rank group (exercises the RCCL / side-stream path on a
        # single GPU; used by the tests)
        self.collectives = self.world > 1 or (force_collectives and dist.is_initialized())
        params = list(params)
        seen, uniq, everything = set(), [], []
        for p in params:
            if id(p) in seen:
                continue
            seen.add(id(p))
            everything.append(p)
            if p.requires_grad:
                uniq.append(p)
        self.params = uniq
        if not uniq:
            raise ValueError("no trainable parameters")
        if comm_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("comm_dtype: torch.float32 or torch.bfloat16")
        self.comm_dtype = comm_dtype
        self.device = uniq[0].device
        self.on_gpu = self.device.type == "cuda"
        self._main = None
        # adopt_params (round 5, ops.deferred_wgrads): parameters whose .grad is left None by zero_grad() -- AccumulateGrad then
        # ADOPTS the incoming gradient without a kernel, and the hook copies it into the bucket view on the stream the gradient
        # was produced on (the weight-gradient side stream for a deferred convolution).  With .grad bound to the view the
        # engine's `view += grad` would run on the backward's stream, i.e. need the weight gradient joined on the spot.
        self.adopt = {id(p) for p in adopt_params} if self.on_gpu else set()
        cap = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets, cur, n = [], [], 0
        for p in reversed(uniq):                      # gradients become ready roughly in reverse creation order
            if cur and n + p.numel() > cap:
                self.buckets.append(_Bucket(cur, self.device, comm_dtype))
                cur, n = [], 0
            cur.append(p)
            n += p.numel()
        if cur:
            self.buckets.append(_Bucket(cur, self.device, comm_dtype))
        self.where = {}
        for bi, b in enumerate(self.buckets):
            b.index = bi
            for i, p in enumerate(b.params):
                self.where[id(p)] = (b, i)
        self.stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self.expected = None          # ids of params that fire in the final backward (learned on step 1, rank-agreed)
        self.ever_fired = None        # ids of params that receive a gradient at all (learned on step 1, rank-agreed)
        self.order = list(range(len(self.buckets)))     # launch order of the buckets (rank 0's completion order of step 1)
        self.fired_final, self.fired_any = set(), set()
        self._fire_seq = []           # step 1: parameter ids in the order they fired in the final backward
        self.armed = False
        self.next_slot = 0            # position in self.order of the next bucket to launch
        self.use_avg = self.on_gpu and dist.is_initialized() and dist.get_backend(self.group) == "nccl"
        self.measure = measure and self.on_gpu
        # per-bucket exposure (one stream-side wait + one event record per bucket and step): only on request -- the aggregate figure
        # above costs two event records per step (keep the bookkeeping out of the timed region by default)
        self.measure_buckets = bool(measure_buckets) and self.measure
        self._exposed = []            # (event at end of backward compute, event after the side stream was joined)
        self._exposed_buckets = []    # (event at arm(), event at end of backward, [(bucket, event: its reduced gradients final)])
        self._launched_now, self._base_ev = [], None
        if self.world > 1:            # identical starting point on every rank, whatever the callers seeded
            src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
            with torch.no_grad():
                synced = everything + [b for b in buffers if torch.is_tensor(b)]
                for t in synced:
                    dist.broadcast(t.data, src=src, group=self.group)
                # writing through .data does not bump ._version, which c2m_amd.ops keys its packed-weight caches on (Winograd U
                # fragments, K-order rows, also of the FROZEN VGG / flow net): a forward that ran before this constructor would
                # leave ranks != 0 convolving with packs of their pre-broadcast weights
                torch.autograd.graph.increment_version(synced)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in uniq]
        self.zero_grad()

    # ---- step protocol -------------------------------------------------------------------------------------
    def zero_grad(self):
        """Replaces optimizer.zero_grad(): zero the buckets and (re)bind every .grad to its bucket view."""
        for b in self.buckets:
            b.flat.zero_()
            b.work, b.launched, b.ready = None, False, False
            for p, v in zip(b.params, b.views):
                p.grad = None if id(p) in self.adopt else v
        self.fired_final.clear()
        self.fired_any.clear()
        self._fire_seq = []
        self.armed = False
        self.next_slot = 0
        self.overlapped_launches = 0  # buckets that went out from a hook, i.e. while backward was still running

    def arm(self):
        """Call right before the last backward() of the step."""
        self.armed = True
        self.fired_final.clear()
        self._fire_seq = []
        self._main = torch.cuda.current_stream(self.device) if self.on_gpu else None    # the stream backward() is called on
        if self.measure and self.collectives:         # time base of the per-bucket exposure (always before every bucket event)
            self._base_ev = torch.cuda.Event(enable_timing=True)
            self._base_ev.record(torch.cuda.current_stream(self.device))
        self._launched_now = []
        if self.expected is not None:
            for b in self.buckets:
                b.pending = sum(1 for p in b.params if id(p) in self.expected)
                b.ready = b.pending == 0
            self._launch_ready()

    def _on_grad(self, p):
        b, i = self.where[id(p)]
        if p.grad is not b.views[i] and p.grad.data_ptr() != b.views[i].data_ptr():
            # .grad was None (an adopt parameter, or someone reset it -- zero_grad(set_to_none=True)): move the gradient into the
            # bucket, on the stream that produced it
            s = ops.deferred_grad_stream(p.grad) if self.on_gpu else None
            if s is not None:
                with torch.cuda.stream(s):
                    b.views[i].copy_(p.grad)
            else:
                b.views[i].copy_(p.grad)
            p.grad = b.views[i]
        self.fired_any.add(id(p))
        if not self.armed:
            return
        self.fired_final.add(id(p))
        if self.expected is None:
            self._fire_seq.append(id(p))
            return
        if b.launched:
            # the agreed `expected` set said this parameter does not fire in the final backward, so its bucket went out
            # without waiting for it: the write above raced with an all-reduce in flight and the other ranks never see
            # this contribution.  Fail loudly (rank-local, like DistributedDataParallel's reduction errors).
            raise RuntimeError("GradientReducer: a parameter outside the set agreed on the first step received a gradient "
                               "in the final backward after its bucket was all-reduced (data-dependent graph); rebuild "
                               "the reducer, or run the first step on data that exercises every branch")
        if id(p) in self.expected:
            b.pending -= 1
            if b.pending == 0:
                b.ready = True
                self._launch_ready()

    def _launch_ready(self):
        while self.next_slot < len(self.order) and self.buckets[self.order[self.next_slot]].ready:
            self._launch(self.buckets[self.order[self.next_slot]])
            self.next_slot += 1
            self.overlapped_launches += 1

    def _collective(self, b):
        """mean over ranks of b.flat, in place (on the current stream)."""
        t = b.flat
        if b.comm is not None:
            b.comm.copy_(t)                       # fp32 -> bf16 (RNE)
            t = b.comm
        if self.use_avg:
            work = dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=self.on_gpu)
        else:
            t.div_(self.world)
            work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=self.on_gpu)
        return work

    def _expand(self, b):
        if b.comm is not None:
            b.flat.copy_(b.comm)

    def _launch(self, b):
        b.launched = True
        if not self.collectives:
            return
        if self.on_gpu:
            # the bucket's gradients were accumulated by backward nodes on the stream of THIS hook -- and, for parameters of the
            # object branch, on its auxiliary stream (ops.aux_branch: AccumulateGrad runs where the parameter was used): wait for all
            cur = torch.cuda.current_stream(self.device)
            self.stream.wait_stream(cur)
            for s in [self._main] + ops.branch_streams(self.device):
                if s is not None and s != cur:
                    self.stream.wait_stream(s)
            with torch.cuda.stream(self.stream):
                b.work = self._collective(b)
                if b.comm is not None:           # stream-ordered behind the collective on the side stream
                    b.work.wait()
                    b.work = None
                    self._expand(b)
                if self.measure_buckets:         # when this bucket's reduced gradients are final (per-bucket exposure)
                    if b.work is not None:
                        b.work.wait()            # (stream-side wait only: orders the event behind the collective)
                        b.work = None
                    b.done_ev = torch.cuda.Event(enable_timing=True)
                    b.done_ev.record(self.stream)
                    self._launched_now.append(b)
        else:
            self._collective(b)
            self._expand(b)

    def _join(self, ev0):
        if self.on_gpu and self.collectives:
            for b in self.buckets:
                if b.work is not None:
                    b.work.wait()
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
            if ev0 is not None:
                ev1 = torch.cuda.Event(enable_timing=True)
                ev1.record(torch.cuda.current_stream(self.device))
                self._exposed.append((ev0, ev1))
                self._exposed_buckets.append((self._base_ev, ev0, [(b.index, b.done_ev) for b in self._launched_now]))
        self._launched_now = []

    def _mark(self):
        if self.measure and self.collectives:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record(torch.cuda.current_stream(self.device))
            return ev0
        return None

    def finish(self):
        """After the last backward: reduce the remaining buckets, join streams, drop never-touched grads."""
        ev0 = self._mark()
        for bi in self.order:                          # the agreed order (see _launch_ready)
            b = self.buckets[bi]
            if not b.launched:
                self._launch(b)
        self.next_slot = len(self.order)
        self._join(ev0)
        if self.expected is None:
            self._agree()
        self._drop_unfired()

    def reduce_all(self):
        """No-overlap form: all-reduce every bucket now (after a HIP-graph replay of forward + backward, whose hooks
        did not run).  Needs the sets learned by one eager step."""
        if self.expected is None:
            raise RuntimeError("GradientReducer.reduce_all: run one eager step (arm / finish) first")
        ev0 = self._mark()
        self._base_ev, self._launched_now = ev0, []
        for b in self.buckets:
            b.work, b.launched = None, False
            self._launch(b)
        self._join(ev0)
        for p in self.params:                          # a replay re-creates no .grad: keep the agreed set bound, drop the rest
            b, i = self.where[id(p)]
            p.grad = b.views[i] if id(p) in self.ever_fired else None

    def _agree(self):
        # Step 1: agree across ranks on which parameters fire (in the final backward / at all) and on the bucket launch
        # order.  Every later step uses the agreed sets, so all ranks issue the same bucket sequence and drop the same
        # gradients even if a data-dependent branch made one rank's graph differ.
        fin = torch.tensor([[float(id(p) in self.fired_final), float(id(p) in self.fired_any)] for p in self.params],
                           device=self.device)
        if self.world > 1:
            dist.all_reduce(fin, op=dist.ReduceOp.MAX, group=self.group)
        fin = fin.cpu()
        self.expected = {id(p) for p, f in zip(self.params, fin) if f[0] > 0}
        self.ever_fired = {id(p) for p, f in zip(self.params, fin) if f[1] > 0}
        # completion position of a bucket on THIS rank = when its last expected parameter fired (never: at the end);
        # buckets without expected parameters are ready at arm() and go first
        pos = {pid: k for k, pid in enumerate(self._fire_seq)}
        late = len(self._fire_seq) + 1
        done = []
        for b in self.buckets:
            exp = [id(p) for p in b.params if id(p) in self.expected]
            done.append(max((pos.get(pid, late) for pid in exp), default=-1))
        order = sorted(range(len(self.buckets)), key=lambda i: (done[i], i))
        t = torch.tensor(order, device=self.device, dtype=torch.int64)
        if self.world > 1:                             # rank 0's order is everybody's order
            dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        self.order = [int(v) for v in t.cpu()]

    def _drop_unfired(self):
        for p in self.params:
            if id(p) not in self.fired_any and id(p) not in self.ever_fired:
                p.grad = None
            elif p.grad is None:                       # an adopt parameter that did not fire on THIS rank in this step: the
                b, i = self.where[id(p)]               # bucket view holds the other ranks' mean (zeros when nobody fired)
                p.grad = b.views[i]

    def exposed_ms(self):
        """Mean GPU time per step between the end of the backward kernels and the moment the compute stream may go on
        (all-reduce time that the overlap did not hide); None unless constructed with measure=True on a GPU."""
        if not self._exposed:
            return None
        torch.cuda.synchronize(self.device)
        return sum(a.elapsed_time(b) for a, b in self._exposed) / len(self._exposed)

    def exposed_ms_per_bucket(self):
        """[(bucket index in launch order, mean ms per step)]: how much later than the end of the backward kernels (and than
        the previous bucket) each bucket's reduced gradients became final -- the share of the exposed all-reduce time every
        bucket is responsible for; 0 for a bucket whose collective was hidden behind backward.  None unless measured."""
        rows = [r for r in self._exposed_buckets if r[0] is not None and r[2]]
        if not rows:
            return None
        torch.cuda.synchronize(self.device)
        acc, cnt = {}, {}
        for base, ev0, evs in rows:
            t_prev = base.elapsed_time(ev0)               # end of backward on the time base
            for k, (bi, ev) in enumerate(evs):
                t = base.elapsed_time(ev)
                acc[(k, bi)] = acc.get((k, bi), 0.0) + max(0.0, t - t_prev)
                cnt[(k, bi)] = cnt.get((k, bi), 0) + 1
                t_prev = max(t_prev, t)
        return [(bi, acc[(k, bi)] / cnt[(k, bi)]) for (k, bi) in sorted(acc)]

    def reset_measurements(self):
        self._exposed = []
        self._exposed_buckets = []

    def bytes_per_step(self):
        """Bytes each rank hands to the all-reduce per step."""
        return (2 if self.comm_dtype == torch.bfloat16 else 4) * sum(b.numel for b in self.buckets)

    def remove(self):
        for h in self._hooks:
            h.remove()
