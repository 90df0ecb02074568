"""event-timed hsp_rev_build (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hs_pose_amd import ops
from hs_pose_amd.ops import _p, _run, _stream
dev = torch.device("cuda:0")
torch.manual_seed(0)
for B, Nq, Ns, k in ((16, 1028, 257, 1), (16, 1028, 64, 1), (16, 1028, 1028, 20), (16, 257, 257, 20)):
    idx = torch.randint(0, Ns, (B, Nq, k), device=dev, dtype=torch.int32)
    off = torch.empty(B, Ns + 1, dtype=torch.int32, device=dev)
    edge = torch.empty(B, Nq * k, dtype=torch.int32, device=dev)
    def run(): _run("hsp_rev_build", (_p(idx), B, Nq, Ns, k, k, _p(off), _p(edge), _stream()))
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): run()
    e1.record(); torch.cuda.synchronize()
    print(f"rev_build B{B} Nq{Nq} Ns{Ns} k{k}: {10 * e0.elapsed_time(e1):.1f} us")
