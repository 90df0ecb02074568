"""micro-driver for profiling: feature-space KNN at the N0 / N1 shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hs_pose_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.relu(torch.randn(16, 1028, 128, device=dev))
y = torch.relu(torch.randn(16, 257, 256, device=dev))
for _ in range(5):
    ops.knn(x, 20); ops.knn(y, 20)
torch.cuda.synchronize()
