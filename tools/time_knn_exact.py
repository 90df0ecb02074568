"""exact-order feature search (eval forward, exact_train): event-timed latency at the stack's shapes (run on the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hs_pose_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for B, N, C, k in [(16, 1028, 128, 20), (16, 257, 128, 20), (16, 257, 256, 20), (16, 64, 256, 8), (1, 1028, 128, 20), (4, 1028, 128, 20)]:
    x = torch.relu(torch.randn(B, N, C, device=dev))
    with ops.exact_scope(True):
        for _ in range(5):
            ops.knn(x, k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            ops.knn(x, k)
        e1.record()
        torch.cuda.synchronize()
    print(f"knn exact B{B} N{N} C{C} k{k}: {1e3 * e0.elapsed_time(e1) / 30:8.1f} us")
