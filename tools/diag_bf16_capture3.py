import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import ref_cpu as oc
from hs_pose_amd import ops, ops_bf16, gcn3d
from hs_pose_amd.config import FLAGS
from hs_pose_amd.FaceRecon import FaceRecon
dev = torch.device("cuda:0")
FLAGS.reset(); FLAGS.train = 0
torch.manual_seed(0)
net = FaceRecon().to(dev).train(); net.set_feature_dtype(torch.bfloat16)
B, N, k, S = 2, 256, 20, 7
xyz = oc.hash_tensor((B, N, 3), 61, 0.05).to(dev)
v = sys.argv[1]
if v == "nowgradf32":
    ops.wgrad = lambda a, b, out=None, colsum=False: out
if v == "nomm":
    real = ops._mm_nn
    ops._mm_nn = lambda g2, W, out=None, alpha=1.0: torch.zeros(g2.shape[0], W.shape[1], device=g2.device)
    ops._mm_nt = lambda x2, W, bias=None, out=None: torch.zeros(x2.shape[0], W.shape[0], device=x2.device)
if v == "nowgradbf":
    ops_bf16._wgrad = lambda A2, B2, out=None, colsum=False: out if out is not None else torch.zeros(A2.shape[1], B2.shape[1], device=A2.device)
if v == "ownmode":
    ops.GEMM_MODE = "own"
if v == "libmode":
    ops.GEMM_MODE = "library"
if v == "fp32":
    net.set_feature_dtype(torch.float32)
    class _D:
        def refresh(self): pass
    net._bf16 = _D()
if v == "stubbwd":
    def bw(ctx, g):
        xyz_, idx_x, arg, F3, arg_o, fg, directions, w_conv23 = ctx.saved_tensors
        return None, None, None, None, torch.zeros_like(directions), torch.zeros(128, 3, 1, device=g.device), torch.zeros_like(w_conv23)
    ops_bf16._SurfaceLayerBf16.backward = staticmethod(bw)
if v == "noste":
    real_bw = ops_bf16._SurfaceLayerBf16.backward
    import types
    orig_float = torch.Tensor.float

if v == "noorl":
    ops_bf16._orl_bwd_accumulate = lambda *a, **k: None
if v == "nocolsum":
    ops_bf16._colsum = lambda x3: torch.zeros(x3.shape[0], x3.shape[2], device=x3.device)
if v == "nogemm":
    real_g = ops.gemm_rows
    def fake(A1, B1, nn1=False, A2=None, B2=None, nn2=False, bias=None, resid=None, cloud_bias=None, rows_per_cloud=0, out=None, alpha=1.0, xyz3=None, w3=None):
        if out is None: out = torch.zeros(A1.shape[0], B1.shape[1] if nn1 else B1.shape[0], dtype=A1.dtype, device=A1.device)
        return out
    ops.gemm_rows = fake
if v == "norun":
    real_run = ops._run
    def fr(name, args, key="", abytes=0, aflops=0):
        if name in ("hsp_rf_surface_bwd_bf16",): return
        return real_run(name, args, key, abytes, aflops)
    ops_bf16._run = fr
if v == "nosteg":
    import hs_pose_amd.ops_bf16 as ob
    src = open(ob.__file__).read()

if "stub" in v:
    def bw(ctx, g):
        xyz_, idx_x, arg, F3, arg_o, fg, directions, w_conv23 = ctx.saved_tensors
        return None, None, None, None, torch.zeros_like(directions), torch.zeros(128, 3, 1, device=g.device), torch.zeros_like(w_conv23)
    ops_bf16._SurfaceLayerBf16.backward = staticmethod(bw)
net._bf16.refresh()
def body():
    if "norefresh" not in v:
        net._bf16.refresh()
    if "nogradnone" not in v:
        for p in net.parameters(): p.grad = None
    with gcn3d.knn_scope():
        o = net.conv_0(xyz, k)
    if "sumbwd" in v:
        o.float().sum().backward()
    else:
        o.backward(torch.ones_like(o))
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): body()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, capture_error_mode="thread_local"):
    body()
gr.replay(); torch.cuda.synchronize()
print(v, "captured + replayed OK")
