"""how the N = 1028 feature search scales with the per-lane list length (k + 1 slots): the insert cost's share of the kernel"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hs_pose_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.relu(torch.randn(16, 1024, 128, device=dev))
for k in (2, 4, 8, 16, 20, 32):
    for _ in range(5):
        ops.knn(x, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.knn(x, k)
    e1.record()
    torch.cuda.synchronize()
    print(f"knn B16 N1024 C128 k{k}: {1e3 * e0.elapsed_time(e1) / 50:8.1f} us")
