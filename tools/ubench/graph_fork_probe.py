"""Probe (run by hand on the GPU box, not a test): does a hipGraph replay run two forked kernel chains side by side?"""
import time

import torch


def main():
    dev = torch.device("cuda:0")
    x = torch.randn(16, 1028, 128, device=dev)
    y = torch.randn(16, 1028, 128, device=dev)
    def chain(t, n=12):
        a = t
        for _ in range(n):
            a = a.sum(dim=1, keepdim=True) * 1e-3 + t      # a reduce + an elementwise: small dependent kernels
        return a
    def seq():
        return chain(x), chain(y)
    side = torch.cuda.Stream()
    def forked():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            b = chain(y)
        a = chain(x)
        cur.wait_stream(side)
        return a, b
    def timeit(fn):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): fn()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): g.replay()
        e1.record(); torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / 50, out
    t1, o1 = timeit(seq)
    t2, o2 = timeit(forked)
    print(f"graph replay: sequential {t1:.1f} us, forked {t2:.1f} us; equal {torch.equal(o1[0], o2[0]) and torch.equal(o1[1], o2[1])}")


if __name__ == "__main__":
    main()
