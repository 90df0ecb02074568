"""CPU, world_size 2, gloo: the N>1 path -- batch sharding and the bucketed gradient mean -- without any
device compute (all-reduced grad == mean of per-rank grads)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from hs_pose_amd.parallel import GradReducer, init_distributed, shard_range
    r, w, dev = init_distributed()
    assert (r, w) == (rank, world) and dev.type == "cpu"
    torch.manual_seed(0)                                   # identical replicas
    model = torch.nn.Sequential(torch.nn.Linear(64, 300), torch.nn.ReLU(), torch.nn.Linear(300, 300),
                                torch.nn.ReLU(), torch.nn.Linear(300, 8))
    unused = torch.nn.Parameter(torch.ones(5))             # never receives a gradient
    params = list(model.parameters()) + [unused]
    red = GradReducer(params, bucket_bytes=100_000)        # several buckets
    assert len(red.buckets) >= 3
    g = torch.Generator().manual_seed(123)
    x = torch.randn(10, 64, generator=g)                   # the GLOBAL batch, split across ranks
    lo, hi = shard_range(10, rank, world)
    for step in range(2):                                  # twice: state resets between steps
        for p in params:
            p.grad = None
        model(x[lo:hi]).pow(2).sum().backward()
        local = [p.grad.clone() for p in model.parameters()]
        red.finish()
        gathered = [[torch.zeros_like(t) for _ in range(world)] for t in local]
        for t, outl in zip(local, gathered):
            dist.all_gather(outl, t)
        for p, outl in zip(model.parameters(), gathered):
            assert torch.allclose(p.grad, sum(outl) / world, rtol=1e-6, atol=1e-7)
        assert torch.equal(unused.grad, torch.zeros(5))
    # mean-of-shards == gradient of the summed loss over the global batch / world
    # (autograd.grad does not accumulate into .grad, so the reducer's hooks stay quiet)
    full = [g_ / world for g_ in torch.autograd.grad(model(x).pow(2).sum(), list(model.parameters()))]
    for p in params:
        p.grad = None
    model(x[lo:hi]).pow(2).sum().backward()
    red.finish()
    for p, f in zip(model.parameters(), full):
        assert torch.allclose(p.grad, f, rtol=1e-4, atol=1e-5)
    red.close()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


def test_grad_reducer_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, "ok"), (1, "ok")]


class _FakeGraphedStep:
    """stands in for hs_pose_amd.graph.GraphedStep on the CPU: the same buffers and entry points, gradients made up"""

    def __init__(self, rank, split):
        self.split = split
        self.flat_grad = torch.zeros(1000)
        self.flat_late, self.flat_early = self.flat_grad[:700], self.flat_grad[700:]
        self.rank, self.step, self.calls = rank, 0, []

    def _grads(self):
        g = torch.Generator().manual_seed(100 * self.step + self.rank)
        return torch.randn(1000, generator=g)

    def run(self):
        self.calls.append("run")
        self.flat_grad.copy_(self._grads())
        self.step += 1

    def run_first(self):
        self.calls.append("first")
        self.flat_late.copy_(self._grads()[:700])

    def run_second(self):
        self.calls.append("second")
        self.flat_early.copy_(self._grads()[700:])
        self.step += 1


def _worker_graphed(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from hs_pose_amd import parallel
    from hs_pose_amd.parallel import graphed_step_with_exchange, init_distributed
    init_distributed()
    assert dist.get_world_size() == world
    issued = []                                            # every collective this rank issues: (op, element count, async?)
    real_all_reduce = dist.all_reduce

    def spy(t, *a, **kw):
        issued.append(("all_reduce", t.numel(), bool(kw.get("async_op", False))))
        return real_all_reduce(t, *a, **kw)
    parallel.dist.all_reduce = spy
    try:
        for split in (False, True):
            fake = _FakeGraphedStep(rank, split)
            for step in range(3):
                graphed_step_with_exchange(fake, world)
                want = sum(torch.randn(1000, generator=torch.Generator().manual_seed(100 * step + r)) for r in range(world)) / world
                assert torch.allclose(fake.flat_grad, want, rtol=1e-6, atol=1e-7), (split, step)
            assert fake.calls == (["first", "second"] * 3 if split else ["run"] * 3)
    finally:
        parallel.dist.all_reduce = real_all_reduce
    # every rank issued the SAME collectives in the SAME order (a rank that skipped or reordered one would hang RCCL, not fail):
    # one flat all-reduce per step unsplit; split: the coarse levels' 700 elements (async, under the second graph), then the rest
    assert issued == [("all_reduce", 1000, False)] * 3 + [("all_reduce", 700, True), ("all_reduce", 300, True)] * 3, issued
    seqs = [None] * world
    dist.all_gather_object(seqs, issued)
    assert all(s_ == issued for s_ in seqs)
    # the training driver's exchange: flat gradient buffers (one per parameter group) averaged in place
    from hs_pose_amd.parallel import mean_flat_gradients
    bufs = [torch.full((50,), float(rank + 1)), torch.arange(7, dtype=torch.float32) * (rank + 1)]
    mean_flat_gradients(bufs)
    mean_rank = (world + 1) / 2.0
    assert torch.allclose(bufs[0], torch.full((50,), mean_rank)) and torch.allclose(bufs[1], torch.arange(7, dtype=torch.float32) * mean_rank)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


@pytest.mark.parametrize("world", [2, 4])
def test_graphed_step_exchange_gloo(world):
    """the gradient exchange bench.py uses around the graph replays (one all-reduce, or two with the first overlapped), on 2 and 4
    ranks against a stub of graph.GraphedStep: flat buffer == mean over the ranks every step, and the sequence of collectives is
    identical on every rank (element counts and order, gathered and compared)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_graphed, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(r, "ok") for r in range(world)]
    assert all(p.exitcode == 0 for p in procs)


def test_shard_range_covers_batch():
    from hs_pose_amd.parallel import shard_range
    for n in (1, 7, 16, 128, 129):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


class _TimedFake(_FakeGraphedStep):
    """a stub step that takes a given time per exchanged step (the probe's clock is wall time)"""

    def __init__(self, rank, split, seconds):
        super().__init__(rank, split)
        self.seconds = seconds

    def run(self):
        import time
        time.sleep(self.seconds)
        super().run()

    def run_second(self):
        import time
        time.sleep(self.seconds)
        super().run_second()


def _worker_choice(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from hs_pose_amd.parallel import choose_exchange_form, init_distributed
    init_distributed()
    got = {}
    # (split seconds, single seconds) per rank: the MAX over the ranks decides, and rank 1 is the slow one in case "b"
    cases = {"a_split_faster": ((0.002, 0.002), (0.02, 0.02)),           # split clearly faster on both ranks
             "b_single_faster_on_max": ((0.002, 0.03), (0.01, 0.01)),    # split faster on rank 0 only: max over ranks says single
             "c_tie_keeps_split": ((0.01, 0.01), (0.01, 0.01))}
    for name, (t_split, t_single) in cases.items():
        forms = {"split": _TimedFake(rank, True, t_split[rank]), "single": _TimedFake(rank, False, t_single[rank])}
        pick, info = choose_exchange_form(forms, world, replays=3)
        assert set(info["probe_ms_per_step"]) == {"split", "single"} and info["replays"] == 3
        # one untimed + three timed exchanged steps of EACH form on every rank, whatever is chosen
        assert forms["split"].calls == ["first", "second"] * 4 and forms["single"].calls == ["run"] * 4
        got[name] = pick
    # forced forms and degenerate inputs never probe (no step of the stub runs)
    f = {"split": _TimedFake(rank, True, 0.0), "single": _TimedFake(rank, False, 0.0)}
    assert choose_exchange_form(f, world, forced="single")[0] == "single" and choose_exchange_form(f, world, forced="split")[0] == "split"
    assert choose_exchange_form({"single": f["single"]}, world)[0] == "single"
    assert choose_exchange_form({"split": f["split"], "single": None}, world, forced="single")[0] == "split"   # forced form absent
    assert choose_exchange_form({}, world)[0] is None
    assert f["split"].calls == [] and f["single"].calls == []
    every = [None] * world
    dist.all_gather_object(every, got)
    assert all(e == got for e in every), every               # all ranks agree (a disagreement would deadlock the collectives)
    assert got["a_split_faster"] == "split" and got["c_tie_keeps_split"] == "split", got
    assert got["b_single_faster_on_max"] == "single", got
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


def test_exchange_form_choice_gloo():
    """the start-up probe bench.py uses when more than one rank runs: both forms timed over the same number of exchanged steps on
    every rank, MAX over the ranks, one decision for all; the overlapped two-graph form is the default (kept on ties), the single
    all-reduce wins only when it is measurably faster; HSP_SPLIT_GRAPH forces a form without probing"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_choice, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, "ok"), (1, "ok")]
    assert all(p.exitcode == 0 for p in procs)


def test_exchange_form_single_rank_needs_no_group():
    from hs_pose_amd.parallel import choose_exchange_form
    a, b = _FakeGraphedStep(0, True), _FakeGraphedStep(0, False)
    pick, info = choose_exchange_form({"split": a, "single": b}, 1)
    assert pick == "single" and a.calls == [] and b.calls == [] and "one rank" in info["reason"]
