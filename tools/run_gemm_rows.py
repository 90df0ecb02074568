"""micro-driver for profiling hsp_gemm_rows: python tools/run_gemm_rows.py M N K nn|nt [bf16] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hs_pose_amd import ops
M, N, K = (int(v) for v in sys.argv[1:4])
nn = sys.argv[4] == "nn"
dt = torch.bfloat16 if "bf16" in sys.argv else torch.float32
dev = torch.device("cuda:0")
A = torch.randn(M, K, device=dev).to(dt)
B = (torch.randn(K, N, device=dev) if nn else torch.randn(N, K, device=dev)).to(dt)
out = torch.empty(M, N, device=dev, dtype=dt)
for _ in range(5):
    ops.gemm_rows(A, B, nn, out=out)
torch.cuda.synchronize()
