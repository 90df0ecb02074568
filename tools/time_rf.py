"""event-timed rf_conv forward / backward at the N0 shape (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hs_pose_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for B, N, C, k, S in ((16, 1028, 128, 20, 7), (16, 257, 256, 20, 7), (16, 64, 512, 8, 7)):
    SC = S * C
    xyz = torch.randn(B, N, 3, device=dev)
    X = torch.relu(torch.randn(B, N, C, device=dev))
    fm = torch.randn(B, N, (S + 1) * C, device=dev)
    dirs = torch.randn(3, SC, device=dev)
    g = torch.randn(B, N, C, device=dev)
    idx = ops.knn(X, k)
    out, arg, fwin = ops._rf_conv_fwd_raw(xyz, idx, dirs, fm, S, True)
    saved = fwin if fwin is not None else fm
    def fwd(): ops._rf_conv_fwd_raw(xyz, idx, dirs, fm, S, True)
    def bwd(): ops._rf_conv_bwd_raw(xyz, idx, dirs, saved, arg, g, S)
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"rf_conv {name} B{B} N{N} C{C}: {20 * e0.elapsed_time(e1):.1f} us")
