"""one deep-K head product, a few launches (for PMC passes): python tools/x3_one.py [N] [K] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hs_pose_amd import ops
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1286
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
M = 16448
g = torch.Generator().manual_seed(0)
A = torch.randn(M, (K + 3) // 4 * 4, generator=g).to(dev)[:, :K]
W = (torch.randn(N, K, generator=g) * 0.05).to(dev)
b = torch.randn(N, generator=g).to(dev)
out = torch.empty(M, N, device=dev)
for _ in range(reps):
    ops.gemm_x3(A, W, False, bias=b, out=out)
torch.cuda.synchronize()
print("done")
