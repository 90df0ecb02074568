"""event-timed latency of the ORL global feature forward (hsp_orl_global_fwd) at the bench step's shapes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hs_pose_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def timed(fn, reps=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for B, N, C, k in [(16, 1028, 128, 20), (16, 257, 128, 20), (16, 257, 256, 20), (16, 64, 256, 8), (4, 1028, 128, 20), (64, 1028, 128, 20)]:
    x = (torch.randn(B, N, 3, generator=g) * 0.05).to(dev)
    f = torch.randn(B, N, C, generator=g).to(dev)
    idx = ops.knn(x, k)
    with torch.no_grad():
        t = timed(lambda: ops.orl_global(f, idx, k))
    print(f"B {B:3d} N {N:5d} C {C:4d} k {k:2d}: {t:7.1f} us (python call included)")
