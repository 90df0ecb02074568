"""device time of ONE hsp_gemm_rows shape (graph of 20 calls): python tools/time_one_gemm.py M N K nn|nt [bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hs_pose_amd import ops
M, N, K = (int(v) for v in sys.argv[1:4])
nn = sys.argv[4] == "nn"
dt = torch.bfloat16 if "bf16" in sys.argv else torch.float32
dev = torch.device("cuda:0")
pad = lambda k: (k + 7) // 8 * 8 if "pitch" in sys.argv else k          # "pitch": rows padded to 16 bytes (K stays ragged)
A = torch.randn(M, pad(K), device=dev).to(dt)[:, :K]
B = (torch.randn(K, N, device=dev) if nn else torch.randn(N, pad(K), device=dev)[:, :K]).to(dt)
if not nn and "pitch" in sys.argv:
    B = torch.randn(N, pad(K), device=dev).to(dt)[:, :K]
out = torch.empty(M, N, device=dev, dtype=torch.float32 if "f32out" in sys.argv else dt)
fn = lambda: ops.gemm_rows(A, B, nn, out=out)
fn(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    fn()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
with torch.cuda.graph(g):
    for _ in range(20):
        fn()
g.replay(); torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    best = min(best, 1e3 * e0.elapsed_time(e1) / 20)
if "lib" in sys.argv:
    fn = lambda: torch.mm(A, B if nn else B.t(), out=out)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"   library: {1e3 * e0.elapsed_time(e1) / 20:8.1f} us")
print(f"M{M} N{N} K{K} {'nn' if nn else 'nt'} {str(dt)[6:]} lib={os.path.basename(os.environ.get('HSP_LIB','libhsp.so'))}: "
      f"{best:8.1f} us  {2.0 * M * N * K / best / 1e6:7.1f} TF")
