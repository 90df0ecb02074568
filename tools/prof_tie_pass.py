"""where one flagged row of the coordinate tie pass spends its time: a private copy of libhsp.so with -DHSP_TIE_PROF (clock64 stamps
around the phases of knn_xyz_ties_kernel) on a cloud with exactly ONE duplicated point.  Run on the GPU box:
    python tools/prof_tie_pass.py          (builds build_tmp/libhsp_prof.so first)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "hs_pose_amd", "csrc")
out = os.path.join(ROOT, "build_tmp", "libhsp_prof.so")
if "--build" in sys.argv or not os.path.exists(out):
    os.makedirs(os.path.dirname(out), exist_ok=True)
    obj = os.path.join(ROOT, "build_tmp", "knn_exact_prof.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-munsafe-fp-atomics", "-I" + os.path.join(ROOT, "include"), "-DHSP_TIE_PROF", "-c",
                           os.path.join(csrc, "knn_exact.hip"), "-o", obj])
    objs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".o") and f != "knn_exact.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj] + objs + ["-o", out])
    if "--build" in sys.argv:
        sys.exit(0)
os.environ["HSP_LIB"] = out
import torch
from hs_pose_amd import ops
from hs_pose_amd._lib import lib
dev = torch.device("cuda:0")
L = lib()
L.hsp_debug_set_tie_prof.argtypes = [ctypes.c_void_p]
prof = torch.zeros(16, dtype=torch.int64, device=dev)
assert L.hsp_debug_set_tie_prof(ctypes.c_void_p(prof.data_ptr())) == 0
for N, k, k2 in ((1028, 20, 4), (257, 20, 4), (64, 8, 0)):
    g = torch.Generator().manual_seed(N)
    x = (torch.rand(1, N, 3, generator=g) * 64).round() / 64 * 0 + torch.randn(1, N, 3, generator=g) * 0.05
    x[0, N // 2] = x[0, 7]                    # one duplicated point: rows 7, N/2 (and rows that hold both among their nearest) tie
    x = x.to(dev)
    for _ in range(3):
        prof.zero_()
        i20, i4 = ops.knn_xyz(x, k, k2)
        torch.cuda.synchronize()
    t = prof.cpu().tolist()
    names = {0: "row start", 8: "fill done", 1: "heap: loaded", 2: "heap: made", 3: "heap: scan done", 4: "heap: sorted", 5: "nth: start",
             6: "nth: done", 7: "sort done"}
    order = sorted((v, s) for s, v in enumerate(t) if v)
    print(f"N={N} k={k}+{k2}: " + "  ".join(f"{names.get(s, s)} +{v - order[0][0]}" for v, s in order) + "  (core clocks, last flagged row)")
