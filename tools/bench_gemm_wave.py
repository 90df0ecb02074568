"""gemm_wave (csrc/gemm_wave.hip) on the dense products of one step (B=16, N=1028): correctness against fp64 and device time per
tile / cut configuration, next to gemm_rows (the LDS-staged kernel) and the BLAS library.   python tools/bench_gemm_wave.py [quick]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hs_pose_amd import ops
from hs_pose_amd._lib import lib

dev = torch.device("cuda:0")
L = lib()


def timeit(fn, reps=10, rounds=3):
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
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / reps)
    return best


def plan(M, N, K1, K2, cfg):
    out = (ctypes.c_int * 10)()
    if not L.hsp_gemm_wave_plan_info(M, N, K1, K2, cfg, out):
        return None
    return list(out)


def cfg_of(rb, ncb, wps, order=0):
    return rb | (ncb << 4) | (wps << 16) | (order << 28)


# name, M, N, K1, nn1, K2, nn2, epilogue
shapes = [
    ("c0.out", 16448, 128, 128, False, 0, False, "rcx"), ("c1.fm", 16448, 1024, 128, True, 0, False, "b"),
    ("c1.out", 16448, 128, 128, False, 128, False, "rc"), ("c2.fm", 4112, 2048, 128, True, 0, False, "b"),
    ("c2.out", 4112, 256, 128, False, 256, False, "rc"), ("c3.fm", 4112, 2048, 256, True, 0, False, "b"),
    ("c3.out", 4112, 256, 256, False, 256, False, "rc"), ("c4.fm", 1024, 4096, 256, True, 0, False, "b"),
    ("c4.out", 1024, 512, 256, False, 512, False, "rc"),
    ("gF.0", 16448, 128, 128, True, 0, False, ""), ("gF.2", 4112, 256, 256, True, 0, False, ""), ("gF.4", 1024, 512, 512, True, 0, False, ""),
    ("gX.1", 16448, 128, 128, True, 1024, False, ""), ("gX.2", 4112, 128, 256, True, 2048, False, ""),
    ("gX.3", 4112, 256, 256, True, 2048, False, ""), ("gX.4", 1024, 256, 512, True, 4096, False, ""),
]
if __name__ != "__main__":
    shapes = []
quick = "quick" in sys.argv
torch.manual_seed(0)
total_best = total_rows = total_lib = 0.0
for name, M, N, K1, nn1, K2, nn2, epi in shapes:
    A1 = torch.randn(M, K1, device=dev)
    B1 = torch.randn(K1, N, device=dev) if nn1 else torch.randn(N, K1, device=dev)
    A2 = torch.randn(M, K2, device=dev) if K2 else None
    B2 = (torch.randn(K2, N, device=dev) if nn2 else torch.randn(N, K2, device=dev)) if K2 else None
    rpc = M // 16
    kw = {}
    if "b" in epi: kw["bias"] = torch.randn(N, device=dev)
    if "r" in epi: kw["resid"] = torch.randn(M, N, device=dev)
    if "c" in epi: kw["cloud_bias"] = torch.randn(16, N, device=dev); kw["rows_per_cloud"] = rpc
    if "x" in epi: kw["xyz3"] = torch.randn(M, 3, device=dev); kw["w3"] = torch.randn(N, 3, device=dev)
    ref = A1.double() @ (B1.double() if nn1 else B1.double().t())
    if K2: ref += A2.double() @ (B2.double() if nn2 else B2.double().t())
    if "b" in epi: ref += kw["bias"].double()
    if "r" in epi: ref += kw["resid"].double()
    if "c" in epi: ref += kw["cloud_bias"].double().repeat_interleave(rpc, 0)[:M]
    if "x" in epi: ref += kw["xyz3"].double() @ kw["w3"].double().t()
    scale = ref.abs().max().item()
    out = torch.empty(M, N, device=dev)
    fl = 2.0 * M * N * (K1 + K2)
    t_rows = timeit(lambda: ops.gemm_rows(A1, B1, nn1, A2, B2, nn2, out=out, **kw))

    def lib_fn():
        torch.mm(A1, B1 if nn1 else B1.t(), out=out)
        if K2: out.addmm_(A2, B2 if nn2 else B2.t())
    t_lib = timeit(lib_fn)
    print(f"== {name}: M{M} N{N} K{K1}+{K2} epi={epi or '-'}  ideal@131TF {fl / 131e6:6.1f} us | gemm_rows {t_rows:7.1f} us {fl / t_rows / 1e6:6.1f} TF | "
          f"library (no epilogue) {t_lib:7.1f} us {fl / t_lib / 1e6:6.1f} TF", flush=True)
    cands = [0]
    for rb, ncb, wps in ((2, 4, 1), (1, 4, 2), (1, 2, 2), (1, 1, 2)):
        for order in ((0, 1) if N > 32 * ncb else (0,)):
            cands.append(cfg_of(rb, ncb, wps, order))
    seen, best = set(), (1e9, None)
    for cfg in cands:
        pl = plan(M, N, K1, K2, cfg)
        if pl is None:
            continue
        key = tuple(pl) + ((cfg >> 28) & 1,)
        if key in seen:
            continue
        seen.add(key)
        rb, ncb, wps, TM, TN, base, nw, urem, npc, _ = pl
        out.fill_(float("nan"))
        ops.gemm_wave(A1, B1, nn1, A2, B2, nn2, out=out, cfg=cfg, **kw)
        torch.cuda.synchronize()
        err = (out.double() - ref).abs().max().item() / scale
        t = timeit(lambda: ops.gemm_wave(A1, B1, nn1, A2, B2, nn2, out=out, cfg=cfg, **kw))
        tol = 2e-6 * max(1.0, ((K1 + K2) / 256.0) ** 0.5)
        flag = "" if err < tol else "   <-- WRONG"
        print(f"   {'auto' if cfg == 0 else 'cfg '} RB{rb} NCB{ncb} wps{wps} order{(cfg >> 28) & 1} tiles {TM}x{TN} per wave {base} blocks left {npc:3d} waves {nw:4d}: "
              f"{t:7.1f} us {fl / t / 1e6:6.1f} TF  err {err:.1e}{flag}", flush=True)
        if err < tol and t < best[0]:
            best = (t, pl)
    print(f"   best {best[0]:7.1f} us  {best[1]}   vs gemm_rows {t_rows:7.1f}  library {t_lib:7.1f}", flush=True)
    total_best += best[0]; total_rows += t_rows; total_lib += t_lib
print(f"TOTAL over shapes: gemm_wave best {total_best:.1f} us, gemm_rows {total_rows:.1f} us, library (no epilogues) {total_lib:.1f} us")
