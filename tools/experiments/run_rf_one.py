"""GPU box: N launches of the receptive-field forward at one shape (for rocprofv3 / PMC passes).
usage: run_rf_one.py <mfma 0|1> [B N C k S surface(0|1) bf16(0|1) reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HSP_RF_MFMA"] = sys.argv[1]
import torch
from hs_pose_amd import ops
from hs_pose_amd.ops import _p, _run, _stream
a = [int(x) for x in sys.argv[2:]] + [None] * 8
B, N, C, k, S, surface, bf, reps = [v if v is not None else d for v, d in zip(a, (16, 1028, 128, 20, 7, 0, 0, 10))]
dev = torch.device("cuda:0")
dt = torch.bfloat16 if bf else torch.float32
g = torch.Generator().manual_seed(1)
xyz = (torch.randn(B, N, 3, generator=g) * 0.05).to(dev)
feat = torch.relu(torch.randn(B, N, 16, generator=g)).to(dev)
idx = ops.knn(xyz if surface else feat, k)
dirs = torch.randn(3, S * C, generator=g).to(dev)
fm = torch.randn(B, N, (S + 1) * C, generator=g).to(dev).to(dt)
out = torch.empty(B, N, C, dtype=dt, device=dev)
arg = torch.empty(B, N, S * C, dtype=torch.uint16, device=dev)
fwin = torch.empty(B, N, S * C, dtype=dt, device=dev)
sfx = "_bf16" if bf else ""
def go():
    if surface:
        _run("hsp_rf_surface_fwd" + sfx, (_p(xyz), _p(idx), _p(dirs), B, N, k, S, C, _p(out), _p(arg), _stream()))
    else:
        _run("hsp_rf_conv_fwd" + sfx, (_p(xyz), _p(idx), _p(dirs), _p(fm), B, N, k, S, C, _p(out), _p(arg), _p(fwin), _stream()))
for _ in range(3): go()
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); go(); e1.record(); torch.cuda.synchronize()
    ts.append(1000 * e0.elapsed_time(e1))
ts.sort()
print(f"mfma={sys.argv[1]} B{B} N{N} C{C} k{k} S{S} surface{surface} bf16{bf}: min {ts[0]:.1f} median {ts[len(ts) // 2]:.1f} max {ts[-1]:.1f} us", flush=True)
