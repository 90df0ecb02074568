"""x3 / library on the head-tower shapes (captured-graph replays); run on the GPU box"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hs_pose_amd import ops
from tools import gemm_tuning
from tools.gemm_gap import timeit
gemm_tuning.enable()
dev = torch.device("cuda:0")
M = 16448
feat = torch.randn(M, 1288, device=dev)[:, :1286]
for N, K, nn, A in ((1024, 1286, False, feat), (256, 1024, False, None), (1286, 1024, True, None), (1024, 256, True, None), (512, 1286, False, feat)):
    A = A if A is not None else torch.randn(M, K, device=dev)
    W = (torch.randn(K, N, device=dev) if nn else torch.randn(N, K, device=dev)) * 0.05
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    fl = 2.0 * M * N * K
    res = {}
    if ops.gemm_x3_ok(A, W, None, None, None if nn else b, None, None, None, out, M, N):
        res["x3"] = timeit(lambda: ops.gemm_x3(A, W, nn, bias=None if nn else b, out=out))
    res["rows"] = timeit(lambda: ops.gemm_rows(A, W, nn, bias=None if nn else b, out=out))
    res["lib"] = timeit(lambda: (torch.mm(A, W, out=out) if nn else torch.addmm(b, A, W.t(), out=out)))
    print(f"M{M} N{N} K{K} {'nn' if nn else 'nt'}: " + "  ".join(f"{k} {v:7.1f} us {fl / v / 1e6:6.1f} TF" for k, v in res.items()), flush=True)
