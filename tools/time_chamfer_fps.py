"""Chamfer distance (forward + backward) and farthest point sampling: kernel time by HIP events on the launch stream,
algorithmic-byte / pair-rate figures, and the C oracle (oracle/hsp_oracle.c, one host thread) on a bounded sample.
Run on the GPU box:  python tools/time_chamfer_fps.py
"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
import torch


def gpu_us(fn, reps=20):
    """device time per call: the call is captured in a hipGraph once and replayed (no host launch overhead in the figure)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def main():
    from hs_pose_amd import ops
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libhsp_oracle.so"))
    dev = torch.device("cuda:0")
    fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32)
    out = []
    for B, n in ((16, 1028), (16, 4096)):
        g = torch.Generator().manual_seed(n)
        a = torch.randn(B, n, 3, generator=g) * 0.1
        b = torch.randn(B, n, 3, generator=g) * 0.1
        ad, bd = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
        t_f = gpu_us(lambda: ops.chamfer(ad.detach(), bd.detach()))

        def fb():
            ad.grad = bd.grad = None
            d1, d2, _, _ = ops.chamfer(ad, bd)
            (d1.sum() + d2.sum()).backward()
        t_fb = gpu_us(fb)
        pairs = 2.0 * B * n * n                              # both directions
        # CPU oracle on a bounded sample (2 clouds), one thread
        s = 2
        an, bn = a[:s].contiguous().numpy(), b[:s].contiguous().numpy()
        d1 = np.empty((s, n), np.float32); d2 = np.empty((s, n), np.float32)
        i1 = np.empty((s, n), np.int32); i2 = np.empty((s, n), np.int32)
        t0 = time.perf_counter()
        lib.hsp_oracle_chamfer_fwd(an.ctypes.data_as(fp), bn.ctypes.data_as(fp), s, n, n, d1.ctypes.data_as(fp),
                                   d2.ctypes.data_as(fp), i1.ctypes.data_as(ip), i2.ctypes.data_as(ip))
        t_cpu = time.perf_counter() - t0
        out.append({"op": "chamfer", "B": B, "n": n, "fwd_us": round(t_f, 1), "fwd_bwd_us": round(t_fb, 1),
                    "fwd_Gpairs_per_s": round(pairs / t_f / 1e3, 1), "clouds_per_s_fwd_bwd": round(B / t_fb * 1e6, 0),
                    "cpu_oracle_fwd_clouds_per_s": round(s / t_cpu, 2), "cpu_threads": 1})
    for B, N, m in ((16, 1028, 256), (16, 4096, 1024), (1, 4096, 1024)):
        g = torch.Generator().manual_seed(N + m)
        p = torch.randn(B, N, 3, generator=g) * 0.1
        pd = p.to(dev)
        t = gpu_us(lambda: ops.fps(pd, m))
        s = min(B, 2)
        pn = p[:s].contiguous().numpy()
        sel = np.empty((s, m), np.int32)
        t0 = time.perf_counter()
        lib.hsp_oracle_fps_f32(pn.ctypes.data_as(fp), s, N, m, sel.ctypes.data_as(ip))
        t_cpu = time.perf_counter() - t0
        out.append({"op": "fps", "B": B, "N": N, "n_samples": m, "us": round(t, 1), "us_per_pick": round(t / m, 3),
                    "clouds_per_s": round(B / t * 1e6, 0), "cpu_oracle_clouds_per_s": round(s / t_cpu, 2), "cpu_threads": 1})
    for r in out:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
