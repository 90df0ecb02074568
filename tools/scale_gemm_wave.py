"""gemm_wave time against problem size (fixed overhead vs rate): M = 16384 k rows, N = 1024, K = 128 / 256 / 512"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hs_pose_amd import ops
from tools.bench_gemm_wave import timeit, cfg_of, dev   # noqa
for K in (128, 256, 512):
    for k in (1, 2, 4):
        M, N = 16384 * k, 1024
        A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); out = torch.empty(M, N, device=dev); bias = torch.randn(N, device=dev)
        fl = 2.0 * M * N * K
        for rb, ncb, wps in ((1, 4, 2), (2, 4, 1)):
            t = timeit(lambda: ops.gemm_wave(A, B, True, out=out, bias=bias, cfg=cfg_of(rb, ncb, wps)))
            print(f"K{K} M{M:7d} RB{rb} NCB{ncb} wps{wps}: {t:8.1f} us {fl / t / 1e6:6.1f} TF", flush=True)
