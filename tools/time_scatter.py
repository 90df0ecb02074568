"""event-timed ORL backward scatter at the N0 shape (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hs_pose_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, N, C, k = 16, 1028, 128, 20
xyz = torch.randn(B, N, 3, device=dev)
F3 = torch.randn(B, N, C, device=dev)
idx = ops.knn(xyz, k)
fg, arg = ops._orl_fwd_raw(F3, idx, k)
gfg = torch.randn(B, C, device=dev)
g = torch.randn(B, N, C, device=dev)
gF = torch.randn(B, N, C, device=dev)
def run():
    ops._orl_bwd_accumulate_raw(gfg, idx, arg, k, gF, extra=g)
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): run()
e1.record(); torch.cuda.synchronize()
print(f"orl bwd scatter B{B} N{N} C{C}: {10 * e0.elapsed_time(e1):.1f} us")
def runf():
    ops._orl_fwd_raw(F3, idx, k)
for _ in range(5): runf()
e0.record()
for _ in range(100): runf()
e1.record(); torch.cuda.synchronize()
print(f"orl fwd: {10 * e0.elapsed_time(e1):.1f} us")
