"""device time of hsp_gemm_x3_f32 on the layer shapes (a graph of 20 calls each; rotating output buffers so the writes reach HBM):
python tools/time_x3.py [M N K1 K2 bias|none]...   (no arguments: the fm / out / input-gradient shapes of the B=16, N=1028 step)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hs_pose_amd import ops

dev = torch.device("cuda:0")
SHAPES = [(16448, 1024, 128, 0, "bias"), (4112, 2048, 128, 0, "bias"), (4112, 2048, 256, 0, "bias"), (1024, 4096, 256, 0, "bias"),
          (4112, 256, 256, 0, "none"), (16448, 128, 128, 0, "none"), (16448, 128, 128, 1024, "none"), (4112, 256, 256, 2048, "none")]
if len(sys.argv) > 1:
    a = sys.argv[1:]
    SHAPES = [(int(a[i]), int(a[i + 1]), int(a[i + 2]), int(a[i + 3]), a[i + 4]) for i in range(0, len(a), 5)]
for (M, N, K1, K2, epi) in SHAPES:
    A1 = torch.randn(M, (K1 + 7) // 8 * 8, device=dev)[:, :K1]          # rows on a 16-byte pitch
    B1 = torch.randn(K1, N, device=dev) * 0.05
    A2 = torch.randn(M, K2, device=dev) if K2 else None
    B2 = torch.randn(N, K2, device=dev) * 0.05 if K2 else None
    bias = torch.randn(N, device=dev) if epi == "bias" else None
    outs = [torch.empty(M, N, device=dev) for _ in range(4)]
    planes = ops.X3Planes()
    with ops.x3_scope(planes):
        fn = lambda i: ops.gemm_x3(A1, B1, True, A2, B2, False, bias=bias, out=outs[i % 4])
        fn(0); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn(0)
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for i in range(20):
                fn(i)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / 20)
    want = A1.double() @ B1.double() + (A2.double() @ B2.double().t() if K2 else 0) + (bias.double() if bias is not None else 0)
    err = ((outs[3].double() - want).abs().max() / want.abs().max()).item()
    fl = 2.0 * M * N * (K1 + K2)
    print(f"M{M} N{N} K{K1}+{K2} {epi:5s} {best:7.1f} us  {fl / best / 1e6:6.1f} TF(fp32-eq)  "
          f"out {4e-6 * M * N / best * 1e3:6.0f} GB/s  rel err {err:.2e}")
