"""the coarse levels' feature-space searches alone (run under rocprofv3 --kernel-trace on the GPU box: tools/trace_one.sh)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hs_pose_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for B, N, C, k in [(16, 257, 128, 20), (16, 257, 256, 20), (16, 64, 256, 8)]:
    x = torch.relu(torch.randn(B, N, C, device=dev))
    for _ in range(20):
        ops.knn(x, k)
    torch.cuda.synchronize()
