"""device time of a few gemm_wave configurations (no correctness check): for ablation builds selected with HSP_LIB"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hs_pose_amd import ops
from tools.bench_gemm_wave import timeit, plan, cfg_of, dev   # noqa

cases = [("c1.fm16k", 16384, 1024, 128, True, 0, False), ("c3.fm4k", 4096, 2048, 256, True, 0, False),
         ("c1.fm", 16448, 1024, 128, True, 0, False), ("c3.fm", 4112, 2048, 256, True, 0, False),
         ("gX.1", 16448, 128, 128, True, 1024, False), ("c1.out", 16448, 128, 128, False, 128, False)]
cfgs = [(1, 4, 2), (1, 4, 1), (2, 4, 1), (1, 2, 2)]
tag = os.path.basename(os.environ.get("HSP_LIB", "libhsp.so"))
for name, M, N, K1, nn1, K2, nn2 in cases:
    A1 = torch.randn(M, K1, device=dev); B1 = torch.randn(K1, N, device=dev) if nn1 else torch.randn(N, K1, device=dev)
    A2 = torch.randn(M, K2, device=dev) if K2 else None
    B2 = torch.randn(N, K2, device=dev) if K2 else None
    out = torch.empty(M, N, device=dev)
    fl = 2.0 * M * N * (K1 + K2)
    for rb, ncb, wps in cfgs:
      for order in (0, 1):
        cfg = cfg_of(rb, ncb, wps, order)
        if plan(M, N, K1, K2, cfg) is None:
            continue
        if order and (N == 128 or os.environ.get("NO_ORDER")):
            continue
        t = timeit(lambda: ops.gemm_wave(A1, B1, nn1, A2, B2, nn2, out=out, cfg=cfg))
        print(f"{tag:16s} {name:7s} RB{rb} NCB{ncb} wps{wps} order{order}: {t:7.1f} us  {fl / t / 1e6:6.1f} TF", flush=True)
