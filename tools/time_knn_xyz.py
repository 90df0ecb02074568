"""event-timed latency of the coordinate search (hsp_knn_xyz_f32: selection + flags, then the tie pass over flagged rows) on the bench
cloud (randn * 0.05, centred: ~1 % of the rows hold an exact tie among their 22 nearest), a tie-free lattice-free cloud and a tiled one"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hs_pose_amd import ops
from hs_pose_amd.ops import _p, _stream, _ws, lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for B, N, k, k2 in [(16, 1028, 20, 4), (16, 257, 20, 4), (16, 64, 8, 0), (4, 1028, 20, 4), (64, 4096, 20, 4)]:
    pc = torch.randn(B, N, 3, generator=g) * 0.05 + torch.tensor([0.0, 0.0, 0.8])
    bench = (pc - pc.mean(dim=1, keepdim=True)).to(dev)
    base = bench[:, : max(N * 2 // 5, 32)]
    tiled = torch.cat([base] * (N // base.shape[1]) + [base[:, : N % base.shape[1]]], dim=1).contiguous()
    for tag, x in (("bench cloud", bench), ("tiled cloud", tiled)):
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        idx = torch.empty(B, N, k, dtype=torch.int32, device=dev)
        idx2 = torch.empty(B, N, max(k2, 1), dtype=torch.int32, device=dev)
        wsb = lib().hsp_knn_xyz_workspace_bytes(B, N)
        ws = _ws(wsb, dev)
        rc = lib().hsp_knn_xyz_f32(_p(x), B, N, k, k2, 1, _p(idx), _p(idx2) if k2 else None, _p(ws), wsb, _p(cnt), _stream())
        assert rc == 0
        torch.cuda.synchronize()
        t_all = timed(lambda: ops.knn_xyz(x, k, k2))
        t_sel = timed(lambda: ops.knn(x, k, _plain_xyz=True))
        print(f"B{B} N{N} k{k}+{k2} {tag}: flagged rows {int(cnt.item())} of {B * N}; selection alone {t_sel:7.1f} us, with the tie pass {t_all:7.1f} us")
