"""gemm_x3 on the deep-K head shapes, event-timed in a loop, and checked against fp64: run twice on the GPU box,
HSP_X3_NO_TALL=1 python tools/time_x3_tall.py   and   python tools/time_x3_tall.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hs_pose_amd import ops
dev = torch.device("cuda:0")
M = 16448
g = torch.Generator().manual_seed(0)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for N, K, nn, with_bias, with_res in ((1024, 1286, False, True, False), (1286, 1024, True, False, True), (1286, 1024, True, False, False),
                                      (512, 1286, False, True, False)):
    Kp = (K + 3) // 4 * 4
    A = torch.randn(M, Kp, generator=g).to(dev)[:, :K]
    W = ((torch.randn(K, N, generator=g) if nn else torch.randn(N, K, generator=g)) * 0.05).to(dev)
    b = torch.randn(N, generator=g).to(dev) if with_bias else None
    Np = (N + 3) // 4 * 4
    res = torch.randn(M, Np, generator=g).to(dev)[:, :N] if with_res else None
    out = torch.empty(M, Np, device=dev)[:, :N]
    fn = lambda: ops.gemm_x3(A, W, nn, bias=b, resid=res, out=out)
    fn()
    ref = A[:2048].double() @ (W.double() if nn else W.double().t())
    if b is not None:
        ref += b.double()
    if res is not None:
        ref += res[:2048].double()
    err = (out[:2048].double() - ref).abs().max().item() / ref.abs().max().item()
    t = timed(fn)
    print(f"M{M} N{N} K{K} {'nn' if nn else 'nt'} bias={with_bias} resid={with_res}: {t:7.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TF-equivalent  "
          f"max err / max |ref| {err:.2e}", flush=True)
