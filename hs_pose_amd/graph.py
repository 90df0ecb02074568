"""hipGraph capture of one HS-stack training step (forward + backward) for launch-bound batch sizes.

At B=16, N=1028 a step is ~350 kernel launches of 5-100 us each; issued eagerly from Python the GPU
idles between them.  ``GraphedStep`` captures zero_grad -> forward -> backward once (static shapes,
static input / gradient buffers) and replays it.  The only host work of the reference's forward -- the
two ``torch.randperm`` draws of the Pool_layers on the CPU default generator (gcn3d.py:243) -- happens
BEFORE each replay, in the same order and on the same generator as the reference, and is uploaded into
static index buffers the captured kernels read.
"""
import torch

from . import gcn3d, ops


def draw_pool_indices(n_points, rate=4, levels=2):
    """consume the CPU default generator exactly like FaceRecon's two Pool_layers (FaceRecon.py:91,96)."""
    out, n = [], n_points
    for _ in range(levels):
        m = int(n / rate)
        out.append(torch.randperm(n)[:m])
        n = m
    return out


class GraphedStep:
    """step = zero_grad; (_, _, feat) = face_recon(centred, obj); feat.backward(dfeat)   as one hipGraph.

    ``centred`` (B,N,3), ``obj`` (B,1), ``dfeat`` (B,N,1286) are static device buffers owned by this object
    (``load_inputs`` copies new data in); parameter ``.grad`` tensors live in the graph's memory pool
    and are overwritten by every replay; ``feat`` is the static output."""

    def __init__(self, face_recon, centred, obj, dfeat, warmup=3, flat_grads=False):
        """flat_grads=True additionally packs every parameter gradient into ONE contiguous buffer
        (``self.flat_grad``, one captured multi-tensor copy per step) so a data-parallel caller can
        all-reduce the step's gradients with a single RCCL collective right after the replay."""
        self.net = face_recon
        self.centred, self.obj, self.dfeat = centred, obj, dfeat
        B, N, _ = centred.shape
        self.n_points = N
        dev = centred.device
        self.pool_idx = [torch.empty(int(N / 4), dtype=torch.int32, device=dev),
                         torch.empty(int(int(N / 4) / 4), dtype=torch.int32, device=dev)]
        self.params = [p for p in face_recon.parameters() if p.requires_grad]
        self.feat = None
        self.flat_grad = None
        if flat_grads:
            self.flat_grad = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev)
        self._upload_pool_indices()
        prev_timer = ops.set_timer(None)               # HIP events cannot be recorded inside a capture
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self._body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._body()
        finally:
            ops.set_timer(prev_timer)

    def _body(self):
        for p in self.params:
            p.grad = None
        with gcn3d.pool_index_feed(self.pool_idx):
            _, _, feat = self.net(self.centred, self.obj)
        feat.backward(self.dfeat)
        self.feat = feat
        if self.flat_grad is not None:
            torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.params],
                      out=self.flat_grad)

    def _upload_pool_indices(self):
        for buf, idx in zip(self.pool_idx, draw_pool_indices(self.n_points)):
            buf.copy_(idx.to(torch.int32), non_blocking=False)

    def load_inputs(self, centred=None, obj=None, dfeat=None):
        for dst, src in ((self.centred, centred), (self.obj, obj), (self.dfeat, dfeat)):
            if src is not None:
                dst.copy_(src, non_blocking=True)

    def run(self):
        """one step: draw + upload the pool indices (host RNG, reference order), replay the graph."""
        self._upload_pool_indices()
        self.graph.replay()
        return self.feat

    def grad_views(self):
        """per-parameter views into flat_grad (after an all-reduce these ARE the averaged gradients)"""
        out, off = [], 0
        for p in self.params:
            out.append(self.flat_grad[off:off + p.numel()].view_as(p))
            off += p.numel()
        return out
