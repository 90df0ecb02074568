"""where workgroup (0, 0) of knn_feat_small_kernel spends its time: a private copy of libhsp.so with -DHSP_KNN_PROF (clock64 stamps per
wave: 0 start, 1 queries staged, 2 MFMA chain done, 3 |x|^2 written, 4 past the barrier, 5 distances in LDS, 6 selected).
Build here (python tools/prof_knn_small.py --build), run on the GPU box."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "hs_pose_amd", "csrc")
out = os.path.join(ROOT, "build_tmp", "libhsp_knnprof.so")
if "--build" in sys.argv or not os.path.exists(out):
    os.makedirs(os.path.dirname(out), exist_ok=True)
    obj = os.path.join(ROOT, "build_tmp", "knn_prof.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-munsafe-fp-atomics", "-I" + os.path.join(ROOT, "include"), "-DHSP_KNN_PROF", "-c",
                           os.path.join(csrc, "knn.hip"), "-o", obj])
    objs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".o") and f != "knn.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj] + objs + ["-o", out])
    if "--build" in sys.argv:
        sys.exit(0)
os.environ["HSP_LIB"] = out
import torch
from hs_pose_amd import ops
from hs_pose_amd._lib import lib
dev = torch.device("cuda:0")
L = lib()
L.hsp_debug_set_knn_prof.argtypes = [ctypes.c_void_p]
prof = torch.zeros(16 * 8, dtype=torch.int64, device=dev)
assert L.hsp_debug_set_knn_prof(ctypes.c_void_p(prof.data_ptr())) == 0
torch.manual_seed(0)
for B, N, C, k in [(16, 257, 128, 20), (16, 257, 256, 20), (16, 64, 256, 8)]:
    x = torch.relu(torch.randn(B, N, C, device=dev))
    for _ in range(3):
        prof.zero_()
        ops.knn(x, k)
        torch.cuda.synchronize()
    t = prof.cpu().view(16, 8).tolist()
    t0 = min(r[0] for r in t if r[0])
    print(f"B={B} N={N} C={C} k={k}  (clock64 ticks since the first wave's start)")
    for w, r in enumerate(t):
        if r[0]:
            print(f"   wave {w:2d}: " + " ".join(f"[{s}]{v - t0:7d}" if v else f"[{s}]      -" for s, v in enumerate(r[:7])))
