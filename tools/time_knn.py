"""event-timed latency of hsp_knn_f32 at the stack's shapes (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hs_pose_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
shapes = [(16, 1024, 128, 20), (16, 1028, 128, 20), (16, 257, 128, 20), (16, 257, 256, 20), (16, 64, 256, 8), (16, 1028, 3, 20), (16, 257, 3, 20)]
for B, N, C, k in shapes:
    x = torch.relu(torch.randn(B, N, C, device=dev)) if C != 3 else torch.randn(B, N, C, device=dev)
    for _ in range(5):
        ops.knn(x, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        ops.knn(x, k)
    e1.record()
    torch.cuda.synchronize()
    print(f"knn B{B} N{N} C{C} k{k}: {1e3 * e0.elapsed_time(e1) / reps:8.1f} us")
