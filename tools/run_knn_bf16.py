"""micro-driver for profiling: feature-space KNN on bf16 rows at the configs[3] shapes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hs_pose_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x = torch.relu(torch.randn(B, 4096, 128, device=dev)).bfloat16()
for _ in range(3):
    ops.knn(x, 20)
torch.cuda.synchronize()
