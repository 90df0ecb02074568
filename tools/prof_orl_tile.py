"""where workgroup 0 of orl_tile_kernel spends its time: a private copy of libhsp.so with -DHSP_ORL_PROF (clock64 stamps of wave 0:
slab filled, then per pass strip written / rows read / pass done).  Run on the GPU box:  python tools/prof_orl_tile.py"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "hs_pose_amd", "csrc")
out = os.path.join(ROOT, "build_tmp", "libhsp_orlprof.so")
if "--build" in sys.argv or not os.path.exists(out):
    os.makedirs(os.path.dirname(out), exist_ok=True)
    obj = os.path.join(ROOT, "build_tmp", "gather_prof.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-munsafe-fp-atomics", "-I" + os.path.join(ROOT, "include"), "-DHSP_ORL_PROF", "-c",
                           os.path.join(csrc, "gather.hip"), "-o", obj])
    objs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".o") and f != "gather.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj] + objs + ["-o", out])
    if "--build" in sys.argv:
        sys.exit(0)
os.environ["HSP_LIB"] = out
import torch
from hs_pose_amd import ops
from hs_pose_amd._lib import lib
dev = torch.device("cuda:0")
L = lib()
L.hsp_debug_set_orl_prof.argtypes = [ctypes.c_void_p]
prof = torch.zeros(48, dtype=torch.int64, device=dev)
assert L.hsp_debug_set_orl_prof(ctypes.c_void_p(prof.data_ptr())) == 0
g = torch.Generator().manual_seed(0)
for B, N, C, k in [(16, 1028, 128, 20), (16, 257, 256, 20), (4, 1028, 128, 20)]:
    x = (torch.randn(B, N, 3, generator=g) * 0.05).to(dev)
    f = torch.randn(B, N, C, generator=g).to(dev)
    idx = ops.knn(x, k)
    with torch.no_grad():
        for _ in range(3):
            prof.zero_()
            ops.orl_global(f, idx, k)
            torch.cuda.synchronize()
    t = prof.cpu().tolist()
    order = [(s, v) for s, v in enumerate(t) if v]
    t0 = order[0][1]
    print(f"B={B} N={N} C={C}: " + " ".join(f"[{s}]+{v - t0}" for s, v in order))
    # the backward (scatter_tile_bwd_kernel, broadcast gradient): zeroed / swept / flushed
    fr = f.clone().requires_grad_(True)
    out = ops.orl_global(fr, idx, k)
    gup = torch.randn_like(out)
    for _ in range(3):
        prof.zero_()
        fr.grad = None
        out.backward(gup, retain_graph=True)
        torch.cuda.synchronize()
    t = prof.cpu().tolist()
    order = [(s, v) for s, v in enumerate(t) if v and s >= 40]
    print("    backward: " + " ".join(f"[{s}]+{v - order[0][1]}" for s, v in order))
