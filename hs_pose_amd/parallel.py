"""Batch-sharded data parallelism for the HS stack: one process per GPU, full parameter replica,
clouds of the global batch split across ranks (every op of the path is per-cloud; BatchNorm statistics
and the Pool_layer randperm draw stay per-rank -- the reference has no SyncBN), and ONE real
exchange per step: the gradient mean over ranks.

The reference has no distributed code at all (engine/train.py:23 hard-codes a single device), so this
is new work whose oracle is "all-reduced grad == mean of the per-rank grads".

xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a ring all-reduce is bound by one link
(~2 * 7/8 * bytes / 153 GB/s): the 38.8 MB of fp32 gradients of the whole model cost ~0.45 ms as one
ring, comparable to the step itself.  Gradients are therefore reduced in a few large buckets (default
8 MiB: 2 for the HS stack, 5 for the whole model) launched from autograd hooks as soon as a bucket's
gradients exist, in reverse registration order (~ backward order), so RCCL overlaps the rest of backward.
"""
import os
from typing import Iterable, List

import torch
import torch.distributed as dist


def init_distributed():
    """(rank, world_size, device) from the torchrun environment; RCCL ('nccl') on GPU, gloo on CPU.

    Two environment hooks let a box with ONE GPU run the N > 1 path through the real kernels (tests/test_gpu_dp_shared.py):
    ``HSP_DIST_BACKEND`` (default ``nccl`` on a GPU, ``gloo`` on the CPU) names the backend and ``HSP_DIST_DEVICE`` (e.g. ``cuda:0``)
    puts every rank on that device instead of ``cuda:LOCAL_RANK``.  RCCL refuses two ranks on one device, so the shared-device form
    goes over gloo, whose collectives on device buffers are staged through the host here (``all_reduce_``)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    force = os.environ.get("HSP_FORCE_DIST", "0") == "1"      # test hook: 1-rank process group
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    device = torch.device(os.environ.get("HSP_DIST_DEVICE") or f"cuda:{local}") if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
    backend = os.environ.get("HSP_DIST_BACKEND") or ("nccl" if use_cuda else "gloo")
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, device


class _Done:
    """the handle of a collective that has already completed (host-staged gloo exchange of a device buffer)"""

    def wait(self):
        return True


def all_reduce_(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
    """``dist.all_reduce`` in place on ``t``.  Over RCCL (and for host tensors) it is exactly that call.  Over gloo with a DEVICE
    buffer -- the shared-GPU test form of ``init_distributed`` -- the buffer goes to the host, is reduced there and comes back,
    synchronously; ``async_op`` then returns a completed handle (no overlap is claimed for that form)."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        host = t.detach().cpu()
        dist.all_reduce(host, op=op, group=group)
        t.copy_(host)
        return _Done() if async_op else None
    return dist.all_reduce(t, op=op, group=group, async_op=async_op)


def describe():
    """what the process group actually is, for the bench line of an N > 1 run: the backend torch.distributed reports ('nccl' IS RCCL
    on ROCm), the rank count IT sees (not the one the command line asked for) and the collective library's version string"""
    if not dist.is_initialized():
        return {"backend": None, "world_size": 1, "collective_library": None}
    lib = None
    if dist.get_backend() == "nccl":
        try:
            v = torch.cuda.nccl.version()
            lib = "rccl " + (".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v))
        except Exception as exc:                              # never lose the line to a version query
            lib = f"rccl (version query failed: {type(exc).__name__})"
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "collective_library": lib}


def shard_range(n_items: int, rank: int, world: int):
    """contiguous [lo, hi) slice of a global batch for this rank (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradReducer:
    """Bucketed, hook-driven gradient mean over the process group.

        reducer = GradReducer(model.parameters())
        loss.backward()          # hooks launch one async all-reduce per completed bucket
        reducer.finish()         # wait, scale by 1/world, write back into p.grad
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 8 << 20, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        order = list(reversed(self.params))                  # ~ the order backward produces gradients
        self.buckets: List[List[torch.nn.Parameter]] = []
        cur, size = [], 0
        for p in order:
            nbytes = p.numel() * p.element_size()
            if cur and size + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {}
        self._flat = []
        for bi, b in enumerate(self.buckets):
            n = sum(p.numel() for p in b)
            self._flat.append(torch.empty(n, dtype=b[0].dtype, device=b[0].device))
            for p in b:
                self._bucket_of[p] = bi
        self._pending = [len(b) for b in self.buckets]
        self._works = [None] * len(self.buckets)
        self._handles = []
        if self.world > 1:
            for p in self.params:
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _on_grad(self, p):
        bi = self._bucket_of[p]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def _launch(self, bi):
        flat, off = self._flat[bi], 0
        for p in self.buckets[bi]:
            n = p.numel()
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            flat[off:off + n].copy_(g.reshape(-1))
            off += n
        self._works[bi] = all_reduce_(flat, group=self.group, async_op=True)

    def finish(self):
        """wait for every bucket, average, scatter back.  Parameters that received no gradient this step
        (unused branches) are reduced as zeros so that all ranks issue identical collectives."""
        if self.world == 1:
            return
        for bi in range(len(self.buckets)):
            if self._works[bi] is None:
                self._launch(bi)
        inv = 1.0 / self.world
        for bi, b in enumerate(self.buckets):
            self._works[bi].wait()
            flat, off = self._flat[bi], 0
            flat.mul_(inv)
            for p in b:
                n = p.numel()
                if p.grad is None:
                    p.grad = flat[off:off + n].view_as(p).clone()
                else:
                    p.grad.copy_(flat[off:off + n].view_as(p))
                off += n
            self._works[bi] = None
        self._pending = [len(b) for b in self.buckets]

    def close(self):
        for h in self._handles:
            h.remove()
        self._handles = []


def graphed_step_with_exchange(graphed, world, group=None):
    """one data-parallel step of a graph-replayed network (hs_pose_amd.graph.GraphedStep): replay, mean the gradients
    over the ranks in the step's flat buffer.

    split form (``graphed.split``): ``run_first()`` leaves the coarse levels' gradients final in ``flat_late`` -- their
    all-reduce is started asynchronously (RCCL's own stream) and runs under ``run_second()``, the fine levels' backward;
    only the second, small exchange is exposed.  Otherwise one all-reduce over ``flat_grad`` after the replay.  Every
    rank issues the same collectives in the same order."""
    if not getattr(graphed, "split", False):
        graphed.run()
        if world > 1 or dist.is_initialized():
            all_reduce_(graphed.flat_grad, group=group)
            graphed.flat_grad.mul_(1.0 / world)
        return
    graphed.run_first()
    late = all_reduce_(graphed.flat_late, group=group, async_op=True)
    graphed.run_second()                                   # overlaps the exchange above
    early = all_reduce_(graphed.flat_early, group=group, async_op=True)
    late.wait()
    early.wait()
    graphed.flat_grad.mul_(1.0 / world)


def mean_flat_gradients(buffers, group=None):
    """gradient mean over the ranks for gradients that already live in flat buffers (the fused optimizer keeps one per
    parameter group, hs_pose_amd/solver.py): one asynchronous all-reduce per buffer, then the 1/world scale.  A no-op
    without a process group or with a single rank."""
    if not dist.is_initialized():
        return
    world = dist.get_world_size(group)
    if world == 1:
        return
    works = [all_reduce_(b, group=group, async_op=True) for b in buffers]
    for w in works:
        w.wait()
    inv = 1.0 / world
    for b in buffers:
        b.mul_(inv)


def choose_exchange_form(forms, world, group=None, replays=3, sync=None, forced=None, probe_single_rank=False):
    """Which gradient-exchange form a data-parallel run uses: ``"split"`` (two graphs, the coarse levels' all-reduce under the
    fine levels' backward -- the default whenever more than one rank runs) or ``"single"`` (one graph, one all-reduce after it).

    ``forms`` maps those names to captured steps (``graph.GraphedStep``-like objects; a form whose capture failed is simply
    absent).  With both present a start-up probe times ``replays`` exchanged steps each way (one untimed step first), takes the
    MAX over the ranks of each time, and rank 0's decision is broadcast so that every rank issues the same collectives: the
    split form stays unless the single form is more than 3 % faster.  ``forced`` ("split" / "single", from HSP_SPLIT_GRAPH=1 / 0)
    skips the probe; ``probe_single_rank`` runs it on a 1-rank group as well (the GPU box's check that the probe's collectives work
    over RCCL).  Returns ``(name, info)`` with ``info`` = what was measured and why, for the bench line."""
    import time
    names = [n for n in ("split", "single") if forms.get(n) is not None]
    if not names:
        return None, {"reason": "no captured form"}
    if forced in names:
        return forced, {"reason": f"forced by HSP_SPLIT_GRAPH ({forced})"}
    if len(names) == 1 or (world == 1 and not probe_single_rank) or not dist.is_initialized():
        pick = names[0] if world > 1 or len(names) == 1 else "single"
        return pick, {"reason": "only one form captured" if len(names) == 1 else "one rank: nothing to overlap"}
    sync = sync or (lambda: None)
    times = {}
    for n in names:                                          # same order on every rank: identical collective sequences
        graphed_step_with_exchange(forms[n], world, group)
        sync(); dist.barrier(group)
        t0 = time.perf_counter()
        for _ in range(replays):
            graphed_step_with_exchange(forms[n], world, group)
        sync()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.barrier(group)
        gathered = [None] * dist.get_world_size(group)
        dist.all_gather_object(gathered, float(t.item()), group=group)
        times[n] = 1e3 * max(gathered) / replays
    decision = ["split" if times["split"] <= 1.03 * times["single"] else "single"]
    dist.broadcast_object_list(decision, src=0, group=group)    # rank 0 decides
    return decision[0], {"reason": "start-up probe (max over ranks, rank 0 decides; split kept unless single is > 3 % faster)",
                         "probe_ms_per_step": {k: round(v, 4) for k, v in times.items()}, "replays": replays}
