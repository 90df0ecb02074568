"""event-timed latency of hsp_wgrad_f32 at the stack's shapes (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hs_pose_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
shapes = [(128, 1024, 16448), (128, 2048, 4112), (256, 2048, 4112), (256, 4096, 1024), (256, 256, 4112), (512, 512, 1024),
          (256, 128, 4112), (128, 128, 16448)]
bf16 = "bf16" in sys.argv
if bf16:
    from hs_pose_amd import ops_bf16
    shapes = [(128, 1024, 262144), (128, 128, 262144), (128, 2048, 65536), (256, 2048, 65536), (256, 4096, 16384), (256, 256, 65536)]
for M, N, K in shapes:
    A = torch.randn(K, M, device=dev); Bm = torch.randn(K, N, device=dev)
    if bf16:
        A, Bm = A.bfloat16(), Bm.bfloat16()
        ops._wgrad_custom = lambda a, b, o, c: ops_bf16._wgrad(a, b, out=o, colsum=c)
    out = torch.empty(M, N, device=dev)
    for colsum in (False,):
        for _ in range(5):
            ops._wgrad_custom(A, Bm, out, colsum)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            ops._wgrad_custom(A, Bm, out, colsum)
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / reps
        print(f"wgrad M{M} N{N} K{K}: {us:8.1f} us  {2e-6 * M * N * K / us:6.1f} TF/s")
