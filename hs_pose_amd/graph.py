"""hipGraph capture of the launch-bound steps of the path.

* ``GraphedStep``       -- one HS-stack training step (zero_grad, forward, backward): what ``bench.py`` times.  At B=16,
  N=1028 it is ~185 launches of 5-100 us; issued eagerly from Python the GPU idles between them.  Optionally packs the
  gradients into one flat buffer for the data-parallel exchange, or splits the step in two graphs so that the first
  all-reduce overlaps the rest of the backward.
* ``GraphedNetwork``    -- ``posenet`` forward / backward of the FULL model as two graphs behind one autograd node, for
  ``engine/train.py``'s step (losses and optimizer stay eager).
* ``GraphedInference``  -- the timed body of ``evaluation/evaluate.py`` (eval forward + generate_RT).
* ``GraphedTrainStep``  -- the whole training step incl. augmentation, losses, backward and the gradient norm as ONE graph
  (+ one fused optimizer launch): what ``bench.py`` times as unit U3.

Static shapes, static input / gradient buffers.  The only host work of the reference's forward -- the two
``torch.randperm`` draws of the Pool_layers on the CPU default generator (gcn3d.py:243) -- happens BEFORE each replay,
in the same order and on the same generator as the reference, and is uploaded into static index buffers the captured
kernels read.
"""
import os

import torch

from . import augment, gcn3d, ops, staging
from .config import FLAGS

# Other threads of the process (RCCL's watchdog polling events, the autograd engine's workers) may call into HIP while
# this thread captures: only this thread's calls are checked against the capture.
_CAPTURE = {"capture_error_mode": "thread_local"}


def draw_pool_indices(n_points, rate=4, levels=2):
    """consume the CPU default generator exactly like FaceRecon's two Pool_layers (FaceRecon.py:91,96)."""
    out, n = [], n_points
    for _ in range(levels):
        m = int(n / rate)
        out.append(torch.randperm(n)[:m])
        n = m
    return out



def alloc_pool_indices(n_points, device):
    """the static index buffers of the two Pool_layers as views of ONE device buffer (so that both are uploaded by one copy)"""
    n1 = int(n_points / 4)
    n2 = int(n1 / 4)
    flat = torch.empty(n1 + n2, dtype=torch.int32, device=device)
    views = [flat[:n1], flat[n1:]]
    views[0]._hsp_flat = flat
    return views


def upload_pool_indices(bufs, n_points):
    """draw the Pool_layer permutations (host generator, reference order) and queue their upload on the current
    stream WITHOUT blocking the host: the copy comes from a ring of pinned staging buffers (hs_pose_amd/staging.py),
    stream-ordered after the previous replay and before the next, so the host can enqueue step i+1 while step i is still
    running.  Buffers made by ``alloc_pool_indices`` travel in ONE copy (a copy kernel per pool cost ~5 us of every step)."""
    draws = draw_pool_indices(n_points)
    flat = getattr(bufs[0], "_hsp_flat", None)
    if flat is not None:
        def fill(pinned):
            o = 0
            for idx in draws:
                pinned[o:o + idx.numel()].copy_(idx)
                o += idx.numel()
        staging.upload(fill, flat.shape, torch.int32, flat.device, out=flat)
        return
    for buf, idx in zip(bufs, draws):
        staging.upload(lambda pinned, idx=idx: pinned.copy_(idx), buf.shape, torch.int32, buf.device, out=buf)


class _preserve_bn_stats:
    """Warm-up forwards run the live network in train mode on the EXAMPLE batch; without this their BatchNorm
    running_mean / running_var / num_batches_tracked updates would leak into the model (and its checkpoints) although
    no training step has happened.  The buffers are snapshotted on entry and restored on exit (after the capture, which
    executes nothing), so a captured object starts from exactly the state the eager path would have."""

    def __init__(self, module):
        self.bufs = [b for m in module.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)
                     for b in (m.running_mean, m.running_var, m.num_batches_tracked) if b is not None]

    def __enter__(self):
        self.saved = [b.detach().clone() for b in self.bufs]
        return self

    def __exit__(self, *exc):
        torch.cuda.synchronize()
        for b, s_ in zip(self.bufs, self.saved):
            b.copy_(s_)
        return False


class GraphedStep:
    """step = zero_grad; (_, _, feat) = face_recon(centred, obj); feat.backward(dfeat)   as one hipGraph.

    ``centred`` (B,N,3), ``obj`` (B,1), ``dfeat`` (B,N,1286) are static device buffers owned by this object
    (``load_inputs`` copies new data in); parameter ``.grad`` tensors live in the graph's memory pool
    and are overwritten by every replay; ``feat`` is the static output."""

    def __init__(self, face_recon, centred, obj, dfeat, warmup=3, flat_grads=False, split=False):
        """flat_grads=True additionally packs every parameter gradient into ONE contiguous buffer
        (``self.flat_grad``, one captured multi-tensor copy per step) so a data-parallel caller can
        all-reduce the step's gradients with a single RCCL collective right after the replay.

        split=True (implies flat_grads) captures the step as TWO graphs cut at ``face_recon.backward_cut``:
        ``run_first()`` = zero_grad, forward and the backward of everything above the cut (conv_3, conv_4, bn3:
        78 % of the gradient bytes, ready after the cheap coarse levels) packed into ``flat_late``; ``run_second()`` =
        the backward of the fine levels packed into ``flat_early``.  The caller starts the all-reduce of ``flat_late``
        between the two, so that exchange overlaps the N=1028 layers' backward."""
        self.net = face_recon
        self.centred, self.obj, self.dfeat = centred, obj, dfeat
        B, N, _ = centred.shape
        self.n_points = N
        dev = centred.device
        self.pool_idx = alloc_pool_indices(N, dev)
        self.params = [p for p in face_recon.parameters() if p.requires_grad]
        self.feat = None
        self.flat_grad = self.flat_late = self.flat_early = None
        self.split = bool(split)
        face_recon.keep_backward_cut = False           # (a split capture of the same network sets it; this object's form decides)
        if flat_grads or split:
            self.flat_grad = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev)
        self._upload_pool_indices()
        prev_timer = ops.set_timer(None)               # HIP events cannot be recorded inside a capture
        keep = _preserve_bn_stats(face_recon).__enter__()
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                if split:
                    self._find_late_params()
                for _ in range(warmup):
                    if split:
                        self._body_first()
                        self._body_second()
                    else:
                        self._body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            if split:
                self.graph2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, **_CAPTURE):
                    self._body_first()
                with torch.cuda.graph(self.graph2, pool=self.graph.pool(), **_CAPTURE):
                    self._body_second()
            else:
                with torch.cuda.graph(self.graph, **_CAPTURE):
                    self._body()
        finally:
            keep.__exit__()
            ops.set_timer(prev_timer)

    def _body(self):
        for p in self.params:
            p.grad = None
        with gcn3d.pool_index_feed(self.pool_idx):
            _, _, feat = self.net(self.centred, self.obj)
        with ops.StepFolds():                           # every p.grad is None: the backward's folds go out in one launch
            feat.backward(self.dfeat)
        self.feat = feat.detach()                       # (keeping the autograd graph alive would pin its accumulators' streams)
        if self.flat_grad is not None:
            torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.params],
                      out=self.flat_grad)

    # ---- the step cut in two (split=True) -------------------------------------------------------------------------
    def _find_late_params(self):
        """late = the parameters the cut tensors do not depend on (their gradients are complete after the first part).
        ``flat_grad`` is laid out [late | early] so each part is one contiguous all-reduce."""
        self.net.keep_backward_cut = True
        with gcn3d.pool_index_feed(self.pool_idx):
            _, _, feat = self.net(self.centred, self.obj)
        cut = list(self.net.backward_cut)
        self.net.backward_cut = None
        below = torch.autograd.grad(cut, self.params, [torch.zeros_like(c) for c in cut], allow_unused=True)
        self.late = [p for p, g in zip(self.params, below) if g is None]
        self.early = [p for p, g in zip(self.params, below) if g is not None]
        del feat, below
        n_late = sum(p.numel() for p in self.late)
        self.flat_late, self.flat_early = self.flat_grad[:n_late], self.flat_grad[n_late:]
        self.params = self.late + self.early           # grad_views() follows the flat layout

    def _body_first(self):
        for p in self.params:
            p.grad = None
        with gcn3d.pool_index_feed(self.pool_idx):
            _, _, feat = self.net(self.centred, self.obj)
        self.feat = feat.detach()
        cut = list(self.net.backward_cut)
        self.net.backward_cut = None
        with ops.StepFolds():
            grads = torch.autograd.grad(feat, cut + self.late, self.dfeat, allow_unused=True)
        self._cut, self._cut_grads = cut, list(grads[:len(cut)])
        torch.cat([(g if g is not None else torch.zeros_like(p)).reshape(-1) for p, g in zip(self.late, grads[len(cut):])],
                  out=self.flat_late)

    def _body_second(self):
        with ops.StepFolds():
            torch.autograd.backward(self._cut, self._cut_grads, inputs=self.early)
        torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.early],
                  out=self.flat_early)
        self._cut = self._cut_grads = None

    def _upload_pool_indices(self):
        upload_pool_indices(self.pool_idx, self.n_points)

    def load_inputs(self, centred=None, obj=None, dfeat=None):
        for dst, src in ((self.centred, centred), (self.obj, obj), (self.dfeat, dfeat)):
            if src is not None:
                dst.copy_(src, non_blocking=True)

    def run(self):
        """one step: draw + upload the pool indices (host RNG, reference order), replay the graph."""
        self._upload_pool_indices()
        self.graph.replay()
        if self.split:
            self.graph2.replay()
        return self.feat

    def run_first(self):
        """split=True: pool indices, forward, backward above the cut -> ``flat_late`` is final"""
        self._upload_pool_indices()
        self.graph.replay()
        return self.feat

    def run_second(self):
        """split=True: the rest of the backward -> ``flat_early`` is final"""
        self.graph2.replay()

    def grad_views(self):
        """per-parameter views into flat_grad (after an all-reduce these ARE the averaged gradients)"""
        out, off = [], 0
        for p in self.params:
            out.append(self.flat_grad[off:off + p.numel()].view_as(p))
            off += p.numel()
        return out


class GraphedTrainStep:
    """One full training step of engine/train.py:76-104 with the device work replayed as a hipGraph:

        zero_grad -> HSPose.forward(do_loss=True) (augmentation, network, 19 losses) -> sum -> backward -> squared
        gradient norm                                              [captured once, replayed]
        optimizer.step() (clip coefficient applied inside) ; scheduler.step()      [one launch + host scalars]

    Host work before each replay, in the reference's order on the CPU default generator: the jitter factors of
    ``defor_3D_pc`` (``torch.rand(PC.shape) * aug_pc_r``) and the two Pool_layer ``randperm`` draws; the augmentation's
    ``torch.rand(..., device=...)`` draws are made inside the graph by the device generator.  ``batch`` is a dict of
    static device tensors with HSPose.forward's keyword names (``load_batch`` copies new data in); gradients land in
    the fused optimizer's flat buffer.  Eagerly the step is CPU-bound (~2500 tiny launches in the losses alone).

    Runtime note.  In round 1 the step still held ATen multi-block reductions (bias-gradient column sums over 16448 rows,
    ``max`` over the points, the ~2 200 small kernels of the losses) and two of them replayed with wrong results under ROCm
    7.2's default AQL-packet capture of graph kernel nodes, so the constructor demanded ``DEBUG_CLR_GRAPH_PACKET_CAPTURE=0``.
    With the heads on ``ops.linear_rows`` / ``ops.points_max`` and the losses in libhsp's five kernels the captured step
    replays equal to the eager step under the DEFAULT runtime (tests/test_gpu_train_graph.py, B=4 N=256 and B=16 N=1028:
    every loss term, gradient and updated parameter), and the flag is no longer needed.  Rounds 2-5 measured ``run()`` at
    30 ms per step at B=16 N=1028 and blamed the runtime's execution of the graph; round 6 found the replay itself takes the
    kernels' 8.0 ms and the rest was HOST time in ``_host_draws``: the jitter draw was scaled by a 49 k-element CPU multiply
    that woke an over-subscribed OpenMP pool (8 ms) and uploaded from pageable memory (a wait for the previous replay).  With
    the draw scaled in serial pieces into the pinned ring, ``run()`` is 8.2 ms -- replay + 0.15 ms -- and this is the form
    ``bench.py`` reports for unit U3.

    Build it BEFORE the network's first eager backward: autograd binds each parameter's gradient accumulator to the
    stream of its first use, and an accumulator bound to another stream than the capture stream is executed outside
    the capture (the replayed graph then reads freed memory).  The warm-up iterations here run on the capture stream."""

    def __init__(self, network, optimizer, batch, scheduler=None, max_norm=5, warmup=3):
        self.net, self.opt, self.sched, self.max_norm = network, optimizer, scheduler, max_norm
        self.batch = batch
        PC = batch["PC"]
        B, N, _ = PC.shape
        self.n_points = N
        dev = PC.device
        self.noise = torch.zeros_like(PC)
        self.pool_idx = alloc_pool_indices(N, dev)
        self.loss_dict, self.total = None, None
        self._host_draws()
        prev_timer = ops.set_timer(None)
        keep = _preserve_bn_stats(network).__enter__()
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self._body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            # capture ON THE WARM-UP STREAM: the parameters' gradient accumulators were bound to it by the warm-up backward
            with torch.cuda.graph(self.graph, stream=side, **_CAPTURE):
                self._body()
        finally:
            keep.__exit__()
            ops.set_timer(prev_timer)

    def _host_draws(self):
        if FLAGS.train:
            # (through the pinned ring, like the pool indices: a copy from pageable memory waits for the previous replay, and the
            # host then cannot queue step i + 1 under step i -- 8 ms of replay + the graph launch's own host time per step, measured
            # 21-41 ms where the replay alone takes 8.1)
            # The scaling runs in pieces below ATen's parallel grain (32768 elements): a 49 k-element multiply wakes the whole
            # OpenMP pool, and with more threads than the container has cores (128 vs a 16-CPU quota on the GPU box) that one
            # multiply cost 8 ms of host time per step.
            r = FLAGS.aug_pc_r

            def fill(pinned):
                src, dst = torch.rand(self.noise.shape).view(-1), pinned.view(-1)
                for o in range(0, src.numel(), 16384):
                    torch.mul(src[o:o + 16384], r, out=dst[o:o + 16384])
            staging.upload(fill, self.noise.shape, torch.float32, self.noise.device, out=self.noise)
        upload_pool_indices(self.pool_idx, self.n_points)

    def _body(self):
        # Gradients are created INSIDE the capture (grad = None first) and then moved into the optimizer's flat buffer
        # with one multi-tensor copy: accumulating straight into the pre-existing flat views makes autograd synchronise
        # the capture stream with the stream those views were made on, and the replayed graph then waits forever.
        params, views = self._params_and_views()
        for p in params:
            p.grad = None
        with gcn3d.pool_index_feed(self.pool_idx), augment.jitter_noise_feed(self.noise):
            _, ld = self.net(do_loss=True, **self.batch)
        total = self.net.total_loss(ld)                     # (the sum over the four sub-dictionaries, engine/train.py:84-90)
        with ops.StepFolds():
            total.backward()
        torch._foreach_copy_(views, [p.grad if p.grad is not None else torch.zeros_like(p) for p in params])
        for p, v in zip(params, views):
            p.grad = v
        self.opt.clip_grad_norm_(self.max_norm)             # device scalar; consumed by the step() outside the graph
        self._gnorm_sq, self._max_norm = self.opt._gnorm_sq, self.opt._max_norm
        self.loss_dict, self.total = ld, total

    def _params_and_views(self):
        params, views = [], []
        for fg in self.opt._flat:
            for p, o in zip(fg.params, fg.offsets):
                params.append(p)
                views.append(fg.view(fg.flat_g, p, o))
        return params, views

    def load_batch(self, batch):
        for k, v in batch.items():
            self.batch[k].copy_(v, non_blocking=True)

    def run(self, check_nan=False):
        """one training step; returns False when ``check_nan`` found a NaN loss (the reference's skip, train.py:91-95)."""
        self._host_draws()
        self.graph.replay()
        if check_nan and bool(torch.isnan(self.total).any()):
            return False
        self.opt._gnorm_sq, self.opt._max_norm = self._gnorm_sq, self._max_norm
        self.opt.step()
        if self.sched is not None:
            self.sched.step()
        return True


class GraphedInference:
    """The timed body of the reference's evaluation loop (evaluation/evaluate.py:90-106) as one hipGraph, for a fixed
    number of instances per call:

        output_dict = network(PC, obj_id, mean_shape, sym)         # eval mode: BatchNorm on running statistics
        pred_RT = generate_RT([p_green_R, p_red_R], [f_green_R, f_red_R], Pred_T, mode='vec', sym)
        pred_s = Pred_s + mean_shape

    ``PC`` (n,N,3), ``obj_id`` (n,) or (n,1), ``mean_shape`` (n,3), ``sym`` (n,4) are static device buffers
    (``load`` copies new data in); ``run()`` draws the two Pool_layer permutations on the host generator like the
    reference's forward does, replays, and returns the static ``(pred_RT (n,4,4), pred_s (n,3), output_dict)``.
    A detector producing a varying instance count keeps one object per count (a capture is a few ms)."""

    def __init__(self, network, PC, obj_id, mean_shape, sym, warmup=2):
        from .geom_utils import generate_RT
        self.net, self._generate_RT = network, generate_RT
        self.PC, self.obj_id, self.mean_shape, self.sym = PC, obj_id, mean_shape, sym
        n, N, _ = PC.shape
        self.n_points = N
        dev = PC.device
        self.pool_idx = alloc_pool_indices(N, dev)
        if network.training:
            raise RuntimeError("GraphedInference: put the network in eval() mode first")
        self._draw()
        prev_timer = ops.set_timer(None)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self._body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, **_CAPTURE):
                self._body()
        finally:
            ops.set_timer(prev_timer)

    def _draw(self):
        upload_pool_indices(self.pool_idx, self.n_points)

    @torch.no_grad()
    def _body(self):
        with gcn3d.pool_index_feed(self.pool_idx):
            out = self.net(PC=self.PC, obj_id=self.obj_id, mean_shape=self.mean_shape, sym=self.sym)
        self.pred_RT = self._generate_RT([out['p_green_R'], out['p_red_R']], [out['f_green_R'], out['f_red_R']],
                                         out['Pred_T'], mode='vec', sym=self.sym)
        self.pred_s = out['Pred_s'] + self.mean_shape
        self.output_dict = out

    def load(self, PC=None, obj_id=None, mean_shape=None, sym=None):
        for dst, src in ((self.PC, PC), (self.obj_id, obj_id), (self.mean_shape, mean_shape), (self.sym, sym)):
            if src is not None:
                dst.copy_(src, non_blocking=True)

    def run(self):
        self._draw()
        self.graph.replay()
        return self.pred_RT, self.pred_s, self.output_dict


class _GraphedNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, runner, anchor):
        ctx.runner = runner
        runner.graph_fwd.replay()
        outs = tuple(o.detach() for o in runner.outs)
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        r = ctx.runner
        for buf, g in zip(r.gouts, gouts):
            if g is None:
                buf.zero_()
            else:
                buf.copy_(g)
        r.graph_bwd.replay()
        have = [(p, g) for p, g in zip(r.params, r.pgrads) if g is not None]
        acc = [(p.grad, g) for p, g in have if p.grad is not None]
        if acc:
            torch._foreach_add_([a for a, _ in acc], [g for _, g in acc])
        for p, g in have:
            if p.grad is None:
                p.grad = g.clone()
        return None, None


class GraphedNetwork:
    """``posenet(PC, obj_id)`` of a training step as two hipGraphs behind ONE autograd node: the forward graph produces
    the ten network outputs in static buffers, the backward graph takes their gradients from static buffers and leaves
    the parameter gradients in static buffers, which the node adds into ``p.grad`` with one multi-tensor add.  Losses,
    augmentation and the optimizer stay eager (they are not part of the captured region), so this is the part of
    ``engine/train.py``'s step that is pure kernels + library GEMMs: ~400 launches and their autograd bookkeeping become
    two replays.  The network's Conv1d(k=1) layers go through ``ops.linear_rows`` (bias gradient by the column-sum
    kernels), BatchNorm through ``ops.bn_relu`` and the max over points through ``ops.points_max``, so the captured
    graphs contain no ATen multi-block reduction (see GraphedTrainStep for why that matters on ROCm 7.2).

    ``runner = GraphedNetwork(net.posenet, PC, obj_id)`` (example inputs of the step's shape, B and N fixed);
    ``outs = runner(PC, obj_id)`` inside the usual step, ``loss.backward()`` as usual.  The returned tensors ALIAS the
    graph's static output buffers: the next call overwrites them -- ``clone()`` anything kept across steps.  Constructing
    the object leaves the BatchNorm running statistics untouched (warm-up runs are rolled back).  The Pool_layer permutations are
    drawn on the host generator before every replay, dropout uses the device generator inside the graph."""

    def __init__(self, posenet, PC, obj_id, warmup=3):
        self.net = posenet
        self.PC = PC.detach().clone()
        self.obj_id = obj_id.detach().clone()
        B, N, _ = PC.shape
        self.n_points = N
        dev = PC.device
        self.pool_idx = alloc_pool_indices(N, dev)
        self.params = [p for p in posenet.parameters() if p.requires_grad]
        self._anchor = torch.zeros(1, device=dev, requires_grad=True)
        upload_pool_indices(self.pool_idx, N)
        prev_timer = ops.set_timer(None)
        keep = _preserve_bn_stats(posenet).__enter__()
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    outs = self._forward()
                    live = [o for o in outs if o is not None and o.requires_grad]
                    torch.autograd.grad(live, self.params, [torch.zeros_like(o) for o in live], allow_unused=True)
                del outs, live
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph_fwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_fwd, **_CAPTURE):
                outs = self._forward()
            self.none_mask = [o is None for o in outs]
            self.outs = [o for o in outs if o is not None]
            self.gouts = [torch.zeros_like(o) for o in self.outs]
            self.graph_bwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_bwd, pool=self.graph_fwd.pool(), **_CAPTURE):
                with ops.StepFolds(bare_wgrad=True):    # (before the .contiguous() copies below read the gradients)
                    self.pgrads = list(torch.autograd.grad(self.outs, self.params, self.gouts, allow_unused=True))
                # contiguous static gradients (copies inside the graph where autograd hands back a transposed view): the
                # multi-tensor add below then takes its fused path instead of one small kernel per parameter
                self.pgrads = [g if g is None or g.is_contiguous() else g.contiguous() for g in self.pgrads]
        finally:
            keep.__exit__()
            ops.set_timer(prev_timer)

    def _forward(self):
        with gcn3d.pool_index_feed(self.pool_idx):
            return self.net(self.PC, self.obj_id)

    def __call__(self, PC, obj_id):
        if PC.shape != self.PC.shape:
            raise RuntimeError(f"GraphedNetwork was captured for {tuple(self.PC.shape)}, got {tuple(PC.shape)}")
        self.PC.copy_(PC)
        self.obj_id.copy_(obj_id.reshape(self.obj_id.shape))
        upload_pool_indices(self.pool_idx, self.n_points)
        live = iter(_GraphedNetFn.apply(self, self._anchor))
        return tuple(None if isnone else next(live) for isnone in self.none_mask)
