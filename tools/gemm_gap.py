"""own vs library, composite by composite: the dense products of every layer of the fp32 step (B=16, N=1028) as the layer nodes
call them (ops._fm_rows / _layer_out_rows / _mm_nn / _grad_in_rows), device time from captured-graph replays.  Run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hs_pose_amd import ops
from tools import library_gemm
_ORIG = {n_: getattr(ops, n_) for n_ in ("_fm_rows", "_layer_out_rows_plain", "_mm_nn", "_mm_nt", "_grad_in_rows", "_tiny_tn", "wgrad",
                                       "_wgrad_ragged_ok", "linear_bn_part_ok", "_layer_out_bn_ok", "_ste_moments_ok", "_thin_wgrad_ok",
                                       "fan_linear_rows_ok", "cloud_cat_linear_ok", "x3_refresh")}

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



if __name__ == "__main__":
    B = 16
    layers = [("conv_1", 1028, 128, 128), ("conv_2", 257, 128, 256), ("conv_3", 257, 256, 256), ("conv_4", 64, 256, 512)]
    S = 7
    tot = {}
    from tools import gemm_tuning
    gemm_tuning.enable()
    rows = []
    for name, N, Cin, C in layers:
        M = B * N
        X = torch.randn(M, Cin, device=dev); W = torch.randn(Cin, (S + 1) * C, device=dev) * 0.05; b = torch.randn((S + 1) * C, device=dev)
        F2 = torch.randn(M, C, device=dev); wste = torch.randn(C, Cin, device=dev) * 0.05; wc2 = torch.randn(C, 2 * C, device=dev) * 0.05
        t2 = torch.randn(B, C, device=dev); out3 = torch.empty(B, N, C, device=dev)
        g2 = torch.randn(M, C, device=dev); gfm = torch.randn(M, (S + 1) * C, device=dev)
        gF = torch.empty(M, C, device=dev); gX = torch.empty(M, Cin, device=dev); fm = torch.empty(M, (S + 1) * C, device=dev)
        Wa = wc2[:, :C]
        comps = [
            ("fm", 2.0 * M * Cin * (S + 1) * C, lambda: ops._fm_rows(X, W, b, out=fm)),
            ("out", 2.0 * M * (Cin + C) * C, lambda: ops._layer_out_rows(X, wste, F2, Wa, t2, out3)),
            ("gF", 2.0 * M * C * C, lambda: ops._mm_nn(g2, Wa, out=gF)),
            ("gX", 2.0 * M * (C + (S + 1) * C) * Cin, lambda: ops._grad_in_rows(g2, wste, gfm, W, gX)),
        ]
        for cname, fl, fn in comps:
            t = {}
            for mode in ("x3", "own", "library"):
                for n_, f_ in _ORIG.items():                 # (the product has no mode switch: the library column swaps the
                    setattr(ops, n_, f_)                     # composites in, tools/library_gemm.py, and back out)
                if mode == "library":
                    library_gemm.enable()
                ops.GEMM_X3 = mode == "x3"
                t[mode] = timeit(fn)
                tot[mode] = tot.get(mode, 0.0) + t[mode]
            print(f"{name} {cname:4s} M{M:6d} Cin{Cin:4d} C{C:4d}  {fl / 1e9:6.2f} GF   x3 {t['x3']:7.1f} us {fl / t['x3'] / 1e6:6.1f} TF   "
                  f"own-f32 {t['own']:7.1f} us {fl / t['own'] / 1e6:6.1f} TF   library {t['library']:7.1f} us {fl / t['library'] / 1e6:6.1f} TF   "
                  f"fp32-MFMA ideal {fl / 155e6:6.1f} us", flush=True)
    print(f"total x3 {tot['x3']:.1f} us   own-f32 {tot['own']:.1f} us   library {tot['library']:.1f} us")
