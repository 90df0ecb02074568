"""device time of hsp_gemm_rows_* against torch (hipBLASLt / rocBLAS) on the shapes of the step; captured-graph replays"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hs_pose_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


shapes = [  # M, N, K, nn
    (16448, 1024, 128, True), (16448, 128, 128, False), (16448, 128, 1024, False), (4112, 2048, 128, True),
    (4112, 2048, 256, True), (4112, 256, 2048, False), (1024, 4096, 256, True), (1024, 256, 4096, False),
    (16448, 1024, 1286, False), (16448, 1286, 1024, True), (16448, 256, 1024, False), (16448, 512, 1286, False)]
dt = torch.bfloat16 if "bf16" in sys.argv else torch.float32
for M, N, K, nn in shapes:
    A = torch.randn(M, K, device=dev).to(dt)
    B = (torch.randn(K, N, device=dev) if nn and dt == torch.float32 else torch.randn(N, K, device=dev)).to(dt)
    nn_ = nn and dt == torch.float32
    out = torch.empty(M, N, device=dev, dtype=dt)
    t_own = timeit(lambda: ops.gemm_rows(A, B, nn_, out=out))
    t_lib = timeit(lambda: torch.mm(A, B if nn_ else B.t(), out=out))
    fl = 2.0 * M * N * K
    print(f"M{M:6d} N{N:5d} K{K:5d} {'nn' if nn_ else 'nt'} {str(dt)[6:]:8s} own {t_own:8.1f} us {fl / t_own / 1e6:7.1f} TF   "
          f"library {t_lib:8.1f} us {fl / t_lib / 1e6:7.1f} TF")
