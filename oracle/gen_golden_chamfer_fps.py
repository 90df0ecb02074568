"""
oracle/gen_golden_chamfer_fps.py -- TEST INFRASTRUCTURE ONLY.  Runs ONLY in the build container.

Pins the two rows whose reference implementation is not the live PyTorch path:

* Chamfer: the reference's own extension, tools/pyTorchChamferDistance/chamfer_distance.cpp, compiled AS IS by
  `make -C oracle ref` into oracle/_ref/cd_ref.so.  Its two *_cuda entry points reference launchers that live in the
  CUDA half (chamfer_distance.cu, unbuildable here); they are left undefined and the module is imported with
  RTLD_LAZY, so only `forward` / `backward` (the CPU path, .cpp:59-87,114-177) are resolved and called -- no stand-in
  code exists.  Every case asserts that oracle/hsp_oracle.c (the restatement the GPU tests check against) returns
  IDENTICAL distances, arg-mins and gradients, then writes tests/golden/chamfer_*.npz.
* FPS: tools/eval_utils.py:107-119 (numpy) on float64 AND float32 clouds -- numpy computes in the dtype it is given --
  including a perturbed-lattice cloud on which a squared-distance rule picks different points than the reference's
  sqrt'ed distances; hsp_oracle_fps_f64 / _f32 must equal the helper on every cloud.  -> tests/golden/fps_*.npz.

usage:  make -C oracle ref && python oracle/gen_golden_chamfer_fps.py
"""
import ctypes
import importlib.util
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path[:0] = [os.path.join(HERE, "stubs"), REF, HERE]

import numpy as np
import torch

import ref_cpu as oc

GOLD = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(1)
clib = ctypes.CDLL(os.path.join(HERE, "libhsp_oracle.so"))


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def save(name, **arrs):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrs)
    mpath = os.path.join(GOLD, "manifest.json")
    man = json.load(open(mpath))
    man["files"][name] = {k: [list(v.shape), str(v.dtype)] for k, v in arrs.items()}
    json.dump(man, open(mpath, "w"), indent=1, sort_keys=True)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


# ------------------------------------------------------------------------------------------------
# Chamfer: the reference extension itself
# ------------------------------------------------------------------------------------------------
so = os.path.join(HERE, "_ref", "cd_ref.so")
assert os.path.exists(so), "run `make -C oracle ref` first"
flags = sys.getdlopenflags()
sys.setdlopenflags(os.RTLD_LAZY)              # the two CUDA launchers stay unresolved: never called
try:
    spec = importlib.util.spec_from_file_location("cd_ref", so)
    cd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cd)
finally:
    sys.setdlopenflags(flags)


print("[chamfer]  reference extension:", so)
for name in oc.CHAMFER_CASES:
    x1, x2, g1, g2 = oc.chamfer_case(name)                         # closed-form inputs the tests re-create
    B, n, m = x1.shape[0], x1.shape[1], x2.shape[1]
    d1, d2 = torch.zeros(B, n), torch.zeros(B, m)                 # caller-allocated, zero-initialised (chamfer_distance.py:19-23)
    i1, i2 = torch.zeros(B, n, dtype=torch.int), torch.zeros(B, m, dtype=torch.int)
    cd.forward(x1, x2, d1, d2, i1, i2)
    gx1, gx2 = torch.zeros_like(x1), torch.zeros_like(x2)
    cd.backward(x1, x2, gx1, gx2, g1, g2, i1, i2)
    # the restatement must be IDENTICAL (same fp32 expression order, strict <, serial scatter order)
    a, b = np.ascontiguousarray(x1.numpy()), np.ascontiguousarray(x2.numpy())
    od1, od2 = np.empty((B, n), np.float32), np.empty((B, m), np.float32)
    oi1, oi2 = np.empty((B, n), np.int32), np.empty((B, m), np.int32)
    clib.hsp_oracle_chamfer_fwd(P(a), P(b), B, n, m, P(od1), P(od2), P(oi1), P(oi2))
    assert np.array_equal(oi1, i1.numpy()) and np.array_equal(oi2, i2.numpy()), name
    assert np.array_equal(od1, d1.numpy()) and np.array_equal(od2, d2.numpy()), name
    ogx1, ogx2 = np.empty_like(a), np.empty_like(b)
    clib.hsp_oracle_chamfer_bwd(P(a), P(b), P(oi1), P(oi2), P(np.ascontiguousarray(g1.numpy())),
                                P(np.ascontiguousarray(g2.numpy())), B, n, m, P(ogx1), P(ogx2))
    assert np.array_equal(ogx1, gx1.numpy()) and np.array_equal(ogx2, gx2.numpy()), name
    # autograd wrapper semantics (chamfer_distance.py:12-55): dist1, dist2 returned, gradients as above
    save(name, dist1=d1.numpy(), dist2=d2.numpy(), idx1=i1.numpy().astype(np.int16), idx2=i2.numpy().astype(np.int16),
         gx1=gx1.numpy(), gx2=gx2.numpy())
    print(f"  {name}: hsp_oracle.c == reference .cpp (dist, idx, grads bit for bit)")

# ------------------------------------------------------------------------------------------------
# FPS: the numpy helper, in both dtypes
# ------------------------------------------------------------------------------------------------
print("[fps]")
from tools.eval_utils import farthest_point_sampling as ref_fps  # needs the cv2 stub


differs = []
for name in oc.FPS_CASES:
    pts, ns = oc.fps_case(name)
    out = {}
    for tag, dt, fn in (("f64", np.float64, clib.hsp_oracle_fps_f64), ("f32", np.float32, clib.hsp_oracle_fps_f32)):
        p = np.ascontiguousarray(pts.astype(dt))
        sel = ref_fps(p, ns)                                       # the reference helper on an array of that dtype
        o = np.empty((1, ns), np.int32)
        fn(P(p), 1, p.shape[0], ns, P(o))
        assert np.array_equal(o[0], sel), (name, tag)
        out["sel_" + tag] = sel.astype(np.int16)
    # what a squared-distance fp32 rule (no sqrt) would have picked: recorded to show the sqrt is load-bearing
    p32 = pts.astype(np.float32)
    dts = np.full(p32.shape[0], np.inf, np.float32)
    cur, sq = 0, []
    for _ in range(ns):
        sq.append(cur)
        d = ((p32 - p32[cur]) ** 2)
        d = (d[:, 0] + d[:, 1]) + d[:, 2]
        dts = np.minimum(dts, d)
        cur = int(np.argmax(dts))
    if not np.array_equal(np.array(sq), out["sel_f32"]):
        differs.append(name)
    out["sel_f32_squared_rule"] = np.array(sq, np.int16)
    if name == "fps_512_64":
        out["sel"] = out["sel_f64"]                                # round-1 key
    save(name, **out)
print("  clouds on which an fp32 squared-distance rule differs from the reference (fp32):", differs)
assert "fps_lattice_512_128" in differs, "the lattice cloud no longer separates the two rules: pick another seed"
print("ok")
