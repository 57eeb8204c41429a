"""N>1 path on CPU: 2 ranks over gloo.  The reducer must deliver mean-of-ranks gradients for a model that is called
directly, with several backward calls per step and parameters that never receive a gradient (the C2M pattern)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from c2m_amd.ddp import GradientReducer


def _retry_rendezvous(fn):
    """A free port found by bind(0) can be taken again before the ranks listen on it, and spawned interpreters import torch
    cold: ONE retry (new port) for failures of the process plumbing.  Numerical assertions are deterministic and fail twice."""
    import functools

    @functools.wraps(fn)
    def run(*a, **k):
        try:
            return fn(*a, **k)
        except Exception:                    # noqa: BLE001
            return fn(*a, **k)
    return run


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(6, 8)
        self.b = nn.Linear(8, 8)
        self.head_g = nn.Linear(8, 3)
        self.head_d = nn.Linear(8, 2)
        self.unused = nn.Linear(4, 4)          # never touched: must end the step with grad None

    def forward(self, x):
        h = torch.tanh(self.b(torch.relu(self.a(x))))
        return self.head_g(h), self.head_d(h.detach()), self.head_d(h)


def _losses(model, x):
    g, d_det, d_att = model(x)
    return d_det.pow(2).mean(), g.abs().mean() + 0.1 * d_att.mean()      # "D" loss, "G" loss (also reaches head_d)


def _single_rank_grads(seed_rank):
    torch.manual_seed(0)
    m = Toy()
    x = torch.randn(5, 6, generator=torch.Generator().manual_seed(100 + seed_rank))
    ld, lg = _losses(m, x)
    ld.backward()
    lg.backward()
    return {k: (p.grad.clone() if p.grad is not None else None) for k, p in m.named_parameters()}


def _worker(rank, world, port, bucket_mb, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        model = Toy()
        red = GradientReducer(list(model.parameters()), bucket_mb=bucket_mb)
        x = torch.randn(5, 6, generator=torch.Generator().manual_seed(100 + rank))
        out = None
        for step in range(3):                  # step 0 learns the fired set, later steps overlap bucket launches
            red.zero_grad()
            ld, lg = _losses(model, x)
            ld.backward()
            red.arm()
            lg.backward()
            red.finish()
            out = {k: (p.grad.clone() if p.grad is not None else None) for k, p in model.named_parameters()}
        q.put((rank, out, len(red.buckets)))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("bucket_mb", [25.0, 0.0002])   # one bucket / several tiny buckets
@_retry_rendezvous
def test_mean_of_rank_gradients_two_ranks(bucket_mb):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bucket_mb, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = [_single_rank_grads(r) for r in range(world)]
    for rank, got, nbuckets in results:
        if bucket_mb < 1:
            assert nbuckets > 2
        for k in got:
            if expect[0][k] is None:
                assert got[k] is None, f"{k} never receives a gradient and must stay None"
                continue
            mean = (expect[0][k] + expect[1][k]) / 2
            torch.testing.assert_close(got[k], mean, rtol=1e-5, atol=1e-7, msg=lambda m: f"rank {rank} {k}: {m}")


def test_single_process_reducer_matches_plain_autograd():
    torch.manual_seed(0)
    model = Toy()
    red = GradientReducer(list(model.parameters()), bucket_mb=0.0002)
    x = torch.randn(5, 6, generator=torch.Generator().manual_seed(100))
    for _ in range(2):
        red.zero_grad()
        ld, lg = _losses(model, x)
        ld.backward()
        red.arm()
        lg.backward()
        red.finish()
    ref = _single_rank_grads(0)
    for k, p in model.named_parameters():
        if ref[k] is None:
            assert p.grad is None
        else:
            torch.testing.assert_close(p.grad, ref[k], rtol=1e-6, atol=1e-8)


class Branchy(nn.Module):
    """head_x only runs when the data asks for it: ranks can end up with different autograd graphs."""

    def __init__(self):
        super().__init__()
        self.a = nn.Linear(6, 8)
        self.head = nn.Linear(8, 3)
        self.head_x = nn.Linear(8, 3)
        self.bn = nn.BatchNorm1d(8)

    def forward(self, x, use_x):
        h = self.bn(torch.relu(self.a(x)))
        out = self.head(h).abs().mean()
        return out + self.head_x(h).pow(2).mean() if use_x else out


def _worker_divergent(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(10 + rank)             # DIFFERENT initial weights and buffers per rank: the reducer must broadcast rank 0's
        model = Branchy()
        with torch.no_grad():
            model.bn.running_mean.fill_(float(rank + 1))
        red = GradientReducer(list(model.parameters()), bucket_mb=0.0002, buffers=list(model.buffers()))
        start = {k: v.clone().numpy() for k, v in model.state_dict().items()}     # numpy: pickled by value
        x = torch.randn(5, 6, generator=torch.Generator().manual_seed(100 + rank))
        grads = []
        for step in range(3):
            red.zero_grad()
            red.arm()
            model(x, use_x=(rank == 0)).backward()      # rank 1 never touches head_x
            red.finish()
            grads.append({k: (p.grad.clone().numpy() if p.grad is not None else None)
                          for k, p in model.named_parameters()})
        q.put((rank, start, grads))
    finally:
        dist.destroy_process_group()


@_retry_rendezvous
def test_rank0_broadcast_and_rank_divergent_graph():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_divergent, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict()
    for _ in range(world):
        rank, start, grads = q.get(timeout=300)
        results[rank] = (start, grads)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0                          # no hang / size mismatch although the graphs differ
    for k in results[0][0]:                             # parameters AND buffers start from rank 0's values
        assert (results[0][0][k] == results[1][0][k]).all(), k
    assert float(results[1][0]["bn.running_mean"][0]) == 1.0
    for step in range(3):
        g0, g1 = results[0][1][step], results[1][1][step]
        for k in g0:                                    # every rank ends every step with the same gradient set and values
            assert (g0[k] is None) == (g1[k] is None), (step, k)
            if g0[k] is not None:
                assert (g0[k] == g1[k]).all(), (step, k)
        assert g1["head_x.weight"] is not None and abs(g1["head_x.weight"]).sum() > 0   # mean with rank 0's contribution


class Backwards(nn.Module):
    """Registration order is the OPPOSITE of what bucketing assumes: the first-registered layer is used last, so the bucket
    that holds it (the last bucket in reverse-registration order) completes FIRST in backward."""

    def __init__(self):
        super().__init__()
        self.last = nn.Linear(8, 3)            # used last in forward -> its gradient fires first
        self.mid = nn.Linear(8, 8)
        self.first = nn.Linear(6, 8)           # used first in forward -> fires last, but sits in bucket 0
        self.frozen = nn.Linear(3, 3)
        for p in self.frozen.parameters():
            p.requires_grad_(False)

    def forward(self, x):
        return self.frozen(self.last(torch.tanh(self.mid(torch.relu(self.first(x)))))).abs().mean()


def _worker_order(rank, world, port, comm_dtype, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(20 + rank)             # per-rank seeding: frozen parameters must be broadcast too
        model = Backwards()
        red = GradientReducer(list(model.parameters()), bucket_mb=0.0002, comm_dtype=comm_dtype)
        frozen = model.frozen.weight.detach().clone().numpy()
        x = torch.randn(5, 6, generator=torch.Generator().manual_seed(100 + rank))
        early = []
        for step in range(3):
            red.zero_grad()
            red.arm()
            loss = model(x)
            # count the buckets that are launched while backward is still running (from the hooks)
            loss.backward()
            early.append(red.overlapped_launches)
            red.finish()
        grads = {k: p.grad.clone().numpy() for k, p in model.named_parameters() if p.grad is not None}
        q.put((rank, frozen, list(red.order), early, grads, red.bytes_per_step(), len(red.buckets)))
    finally:
        dist.destroy_process_group()


def _single_backwards(rank, state):
    m = Backwards()
    m.load_state_dict(state)
    x = torch.randn(5, 6, generator=torch.Generator().manual_seed(100 + rank))
    m(x).backward()
    return {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("comm_dtype", [torch.float32, torch.bfloat16])
@_retry_rendezvous
def test_out_of_order_readiness_agreed_launch_order_and_bf16_buckets(comm_dtype):
    """(1) buckets launch as they complete, in the order rank 0 saw on the first step -- not in bucket-index order, where the
    late bucket 0 would hold back all others until finish(); (2) frozen parameters are broadcast from rank 0 as well;
    (3) bf16 buckets: half the bytes on the wire, mean-of-ranks gradient within bf16 rounding."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_order, args=(r, world, port, comm_dtype, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=300)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (fz0, order0, early0, g0, bytes0, nb), (fz1, order1, early1, g1, _, _) = res[0], res[1]
    assert (fz0 == fz1).all(), "frozen parameters must start from rank 0's values on every rank"
    assert order0 == order1 and sorted(order0) == list(range(nb)) and nb >= 4
    assert order0 != list(range(nb)), "bucket 0 holds the parameters that fire last: it must not be first in the launch order"
    assert order0[-1] in (0, 1) and order0[0] in (nb - 1, nb - 2)      # `first`'s weight / bias fire last, `last`'s first
    assert early0[0] == 0 and early0[1] >= nb - 1 and early0[2] >= nb - 1, early0     # step 1 learns; later steps overlap
    torch.manual_seed(20)
    state = Backwards().state_dict()             # rank 0's initial weights
    e0, e1 = _single_backwards(0, state), _single_backwards(1, state)
    total = sum(p.numel() for p in Backwards().parameters() if p.requires_grad)
    assert bytes0 == total * (2 if comm_dtype == torch.bfloat16 else 4)
    for k in e0:
        mean = (e0[k] + e1[k]) / 2
        assert (g0[k] == g1[k]).all(), k
        tol = dict(rtol=1e-2, atol=1e-3) if comm_dtype == torch.bfloat16 else dict(rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(torch.from_numpy(g0[k]), mean, **tol, msg=lambda m: f"{k}: {m}")


def test_unexpected_late_gradient_fails_loudly():
    """A parameter outside the agreed set that fires in the final backward after its bucket went out must raise, not race
    with the all-reduce in flight."""
    torch.manual_seed(0)
    model = Branchy()
    red = GradientReducer(list(model.parameters()), bucket_mb=1e-6)       # one bucket per parameter
    x = torch.randn(5, 6)
    red.zero_grad()
    red.arm()
    model(x, use_x=False).backward()              # step 1: head_x never fires -> not expected
    red.finish()
    red.zero_grad()
    red.arm()                                      # head_x's buckets have nothing to wait for: launched right here
    with pytest.raises(RuntimeError, match="outside the set agreed"):
        model(x, use_x=True).backward()
