import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import ref_cpu as oc
from hs_pose_amd import ops, ops_bf16, gcn3d
from hs_pose_amd._lib import lib
from hs_pose_amd.config import FLAGS
from hs_pose_amd.FaceRecon import FaceRecon
dev = torch.device("cuda:0")
FLAGS.reset(); FLAGS.train = 0
torch.manual_seed(0)
net = FaceRecon().to(dev).train(); net.set_feature_dtype(torch.bfloat16)
net._bf16.refresh()
B, N, k, S, C = 2, 256, 20, 7, 128
xyz = oc.hash_tensor((B, N, 3), 61, 0.05).to(dev)
idx = ops.knn(xyz, k)
g = oc.hash_tensor((B, N, C), 1, 1.0).to(dev).bfloat16()
F3 = oc.hash_tensor((B, N, C), 2, 1.0).to(dev).bfloat16()
fg, arg_o = ops_bf16._orl_fwd(F3, idx, k)
c0 = net.conv_0
w_conv2 = c0.conv2.weight.squeeze(-1)
_, c2T = ops_bf16.copies_of(w_conv2)
g2, F2, x2 = g.view(B * N, C), F3.view(B * N, C), xyz.view(B * N, 3)
gt = ops_bf16._colsum(g)
gF3 = torch.zeros(B, N, C, dtype=torch.bfloat16, device=dev)
gfg = torch.zeros(B, C, device=dev)
arg = torch.zeros(B, N, S * C, dtype=torch.uint16, device=dev)
g_conv2 = torch.empty_like(w_conv2)
which = sys.argv[1]
def body():
    if which == "colsum": ops_bf16._colsum(g)
    elif which == "wgradbf": ops_bf16._wgrad(g2, F2, out=g_conv2[:, :C])
    elif which == "wgradf32": ops.wgrad(gt, fg, out=g_conv2[:, C:])
    elif which == "gemm": ops.gemm_rows(g2, c2T[:C], out=gF3.view(B * N, C))
    elif which == "mmnn": ops._mm_nn(gt, w_conv2[:, C:], alpha=1.0 / N)
    elif which == "orlbwd": ops_bf16._orl_bwd_accumulate(gfg, idx, arg_o, gF3, g)
    elif which == "surfbwd":
        gD = torch.empty_like(c0.directions)
        wsb = lib().hsp_rf_bwd_scatter_workspace_bytes(B, S * C); ws = ops._ws(wsb, dev)
        ops._run("hsp_rf_surface_bwd_bf16", (ops._p(xyz), ops._p(c0.directions), ops._p(arg), ops._p(gF3), B, N, S, C, ops._p(gD), ops._p(ws), wsb, ops._stream()))
    elif which == "ste": return g2.float().t() @ x2
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): body()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, capture_error_mode="thread_local"):
    body()
gr.replay(); torch.cuda.synchronize()
print(which, "captured + replayed OK")
