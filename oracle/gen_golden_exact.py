"""
oracle/gen_golden_exact.py -- TEST INFRASTRUCTURE ONLY.  Runs ONLY in the build container.

Pins the ROUNDING ORDER of the reference's CPU forward: imports the reference from /root/reference
(stubs under oracle/stubs/), one torch thread, and writes tests/golden/exact_*.npz:

  exact_statements   -- the statements the device kernels restate, each ASSERTED here against the imported reference's own ops
                        before anything is written: its GEMMs (matmul and Conv1d(k=1)) are k-ordered fp32 fma chains from 0 for
                        K <= 256 and a sum of 256-wide block chains above; mean over the supports = sequential sum / S; mean over
                        the points = ATen's 16-row cascade / N; F.normalize's norm = fma chain; eval BatchNorm =
                        ((x - m) * invstd) * w + b with invstd from at::sqrt (MKL VML: NOT correctly rounded -- the fixture records
                        how many channels differ from the IEEE value).  Stores small closed-form cases with the reference's outputs.
  exact_layers       -- reference HSlayer_surface / HS_layer (eval) on closed-form inputs: full outputs (N = 96...128 points).
  exact_stack_1028   -- the reference-initialised FaceRecon of stack_refinit_eval_1028 (same seed, same cloud): strided samples of
                        fm_0 ... fm_4 and feat.  A forward whose feature rows carry the reference's bits reproduces them exactly.
The reference's source never enters this repo: fixtures hold numbers only.

usage:  python oracle/gen_golden_exact.py [--debug-dump build_tmp/exact_debug.npz]
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path[:0] = [os.path.join(HERE, "stubs"), REF, HERE]

import numpy as np
import torch
import torch.nn.functional as F

import config.config  # noqa: F401  (reference flag definitions)
from absl import flags

FLAGS = flags.FLAGS
import network.fs_net_repo.gcn3d as rg
from network.fs_net_repo.FaceRecon import FaceRecon as RefFaceRecon

import ref_cpu as oc

GOLD = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(1)
f32, f64 = np.float32, np.float64


def chain(x, w, acc=None):
    """k-ordered fp32 fma chain from 0 (float64 holds a*b exactly; the double rounding of a*b + c is a 2^-29 event)"""
    M, K = x.shape
    acc = np.zeros((M, w.shape[1]), f32) if acc is None else acc
    x64, w64 = x.astype(f64), w.astype(f64)
    for k in range(K):
        acc = (x64[:, k:k + 1] * w64[k:k + 1, :] + acc.astype(f64)).astype(f32)
    return acc


def block_chains(x, w, kb=256):
    tot = None
    for k0 in range(0, x.shape[1], kb):
        c = chain(x[:, k0:k0 + kb], w[k0:k0 + kb])
        tot = c if tot is None else (tot + c).astype(f32)
    return tot


def cascade_mean(x):
    """ATen's multi_row_sum over dim 1 of (B,N,C), then / N"""
    B, N, C = x.shape
    lp = max(4, int(np.ceil(np.log2(N))) // 4) if N > 1 else 4
    step, mask = 1 << lp, (1 << lp) - 1
    a = [np.zeros((B, C), f32) for _ in range(4)]
    i = 0
    while i + step <= N:
        for _ in range(step):
            a[0] = (a[0] + x[:, i]).astype(f32); i += 1
        for j in range(1, 4):
            a[j] = (a[j] + a[j - 1]).astype(f32); a[j - 1] = np.zeros((B, C), f32)
            if (i & (mask << (j * lp))) != 0:
                break
    while i < N:
        a[0] = (a[0] + x[:, i]).astype(f32); i += 1
    for j in range(1, 4):
        a[0] = (a[0] + a[j]).astype(f32)
    return (a[0] / f32(N)).astype(f32)


def statements():
    out = {}
    # --- products -------------------------------------------------------------------------------------------------------------
    for tag, (M, K, N) in {"k128": (96, 128, 64), "k256": (96, 256, 64)}.items():
        X = torch.relu(oc.hash_tensor((M, K), 7001 + K, 1.0)); W = oc.hash_tensor((K, N), 7002 + K, 0.05); b = oc.hash_tensor((N,), 7003, 0.1)
        ref = (X.view(2, M // 2, K) @ W + b).view(M, N)                            # gcn3d.py:171
        want = (chain(X.numpy(), W.numpy()) + b.numpy()).astype(f32)
        assert np.array_equal(ref.numpy(), want), tag
        Wt = oc.hash_tensor((N, K), 7004 + K, 0.05)
        refc = oc._conv1x1(X.view(2, M // 2, K), Wt.unsqueeze(-1)).reshape(M, N)    # gcn3d.py:149
        assert np.array_equal(refc.numpy(), chain(X.numpy(), Wt.t().contiguous().numpy())), tag + " conv1d"
        out[f"mm_{tag}"] = ref.numpy(); out[f"conv_{tag}"] = refc.numpy()
    # conv2 over cat[F, f_global]: 2C = 256 one chain, 2C = 512 two block chains (gcn3d.py:111,185)
    for C in (128, 256):
        Fm = oc.hash_tensor((2, 48, C), 7100 + C, 1.0); fg = oc.hash_tensor((2, 1, C), 7101 + C, 0.5).repeat(1, 48, 1)
        W2 = oc.hash_tensor((C, 2 * C, 1), 7102 + C, 0.05)
        ref = oc._conv1x1(torch.cat([Fm, fg], -1), W2).reshape(96, C)
        cat = torch.cat([Fm, fg], -1).reshape(96, 2 * C).numpy()
        want = block_chains(cat, W2.squeeze(-1).t().contiguous().numpy())
        assert np.array_equal(ref.numpy(), want), f"conv2 C{C}"
        out[f"conv2_c{C}"] = ref.numpy()
    # --- reductions -----------------------------------------------------------------------------------------------------------
    x = oc.hash_tensor((2, 40, 7, 32), 7200, 1.0)
    ref = x.mean(dim=2).numpy()
    acc = x.numpy()[:, :, 0].copy()
    for s in range(1, 7):
        acc = (acc + x.numpy()[:, :, s]).astype(f32)
    assert np.array_equal(ref, (acc / f32(7)).astype(f32))
    for N in (1028, 257, 64, 300):
        x = oc.hash_tensor((2, N, 32), 7300 + N, 1.0)
        assert np.array_equal(x.mean(dim=1).numpy(), cascade_mean(x.numpy())), N
        out[f"mean_n{N}"] = x.mean(dim=1).numpy()
    # --- |x|^2 of a transposed view (what conv_3 gets, FaceRecon.py:94-95) and the cloud's mean (PoseNet9D.py:25): outer sums
    for (N, C) in ((257, 256), (100, 64), (1028, 32)):
        t = torch.relu(oc.hash_tensor((2, C, N), 7350 + N, 1.0))               # (B,C,N) as BatchNorm1d leaves it
        out[f"quad_outer_n{N}"] = torch.sum(t.transpose(1, 2) ** 2, dim=2).numpy()
    for N in (1028, 100):
        pts = oc.hash_tensor((2, N, 3), 7360 + N, 0.05); pts[:, :, 2] += 0.8
        m = pts.mean(dim=1, keepdim=True)
        out[f"centre_mean_n{N}"] = m.numpy(); out[f"centre_local_n{N}"] = (pts - m).numpy()
    # --- F.normalize ------------------------------------------------------------------------------------------------------------
    v = oc.hash_tensor((500, 3), 7400, 0.05)
    a, b_, c = v.numpy().T
    n2 = (c.astype(f64) * c.astype(f64) + (b_.astype(f64) * b_.astype(f64) + (a * a).astype(f32).astype(f64)).astype(f32).astype(f64)).astype(f32)
    nr = np.maximum(np.sqrt(n2), f32(1e-12))
    want = np.stack([a / nr, b_ / nr, c / nr], -1).astype(f32)
    assert np.array_equal(F.normalize(v, dim=-1).numpy(), want)
    assert np.array_equal(F.normalize(v.t().contiguous(), dim=0).numpy(), want.T)
    out["normalize"] = want
    # --- eval BatchNorm -----------------------------------------------------------------------------------------------------------
    C = 256
    rm, rv = oc.hash_tensor((C,), 7500, 1.0), oc.hash_tensor((C,), 7501, 1.2).abs() + 0.2
    w, bb = oc.hash_tensor((C,), 7502, 1.0), oc.hash_tensor((C,), 7503, 1.0)
    x = oc.hash_tensor((2, 60, C), 7504, 1.5)
    ref = F.batch_norm(x.transpose(1, 2), rm, rv, w, bb, False, 0.1, 1e-5).transpose(1, 2).contiguous()
    inv = 1 / torch.sqrt(rv + 1e-5)                                         # (ATen's own: MKL VML sqrt)
    want = ((((x.numpy() - rm.numpy()).astype(f32) * inv.numpy()).astype(f32) * w.numpy()).astype(f32) + bb.numpy()).astype(f32)
    assert np.array_equal(ref.numpy(), want)
    inv_ieee = (f32(1) / np.sqrt((rv.numpy() + f32(1e-5)).astype(f32))).astype(f32)
    out["bn_eval"] = ref.numpy(); out["bn_invstd"] = inv.numpy()
    out["bn_invstd_differs_from_ieee"] = np.array([int((inv.numpy() != inv_ieee).sum()), C])
    np.savez_compressed(os.path.join(GOLD, "exact_statements.npz"), **out)
    print("exact_statements: every statement equals the reference's op; invstd differs from the IEEE value in",
          out["bn_invstd_differs_from_ieee"].tolist(), "channels")


def topk_ties():
    """torch.topk(largest=False) on rows full of exactly equal values: the order libstdc++ leaves them in (both branches of
    ATen's TopKImpl.h: nth_element + sort for m * 64 > N, partial_sort otherwise)"""
    out = {}
    g = torch.Generator().manual_seed(5)
    for tag, (R, N, m, q) in {"n1028_m21": (24, 1028, 21, 64), "n257_m21": (24, 257, 21, 32), "n64_m9": (24, 64, 9, 16),
                              "n1028_m5": (24, 1028, 5, 64), "n4096_m21": (8, 4096, 21, 128)}.items():
        d = torch.randint(0, q * 8, (R, N), generator=g).float() / q
        out["d_" + tag] = (d * q).to(torch.int16).numpy()                   # (stored as the integer numerators)
        out["q_" + tag] = np.array([q, m], np.int32)
        out["i_" + tag] = torch.topk(d, m, dim=-1, largest=False, sorted=True)[1].numpy().astype(np.int16)
    np.savez_compressed(os.path.join(GOLD, "exact_topk_ties.npz"), **out)
    print("exact_topk_ties:", [k_ for k_ in out if k_.startswith("i_")])


def layers():
    out = {}
    S, k = 7, 20
    xyz = oc.hash_tensor((2, 128, 3), 7600, 0.05)
    # HSlayer_surface
    m = rg.HSlayer_surface(kernel_num=128, support_num=S).eval()
    sd = m.state_dict()
    for i, (kk, v) in enumerate(sd.items()):
        oc.hash_fill_(v, 7610 + i, 0.3 if "STE" in kk else 0.05)
    with torch.no_grad():
        out["surface"] = m(xyz, k).numpy()
    for tag, (Cin, Co, n, kk_) in {"hs128": (128, 128, 128, 20), "hs256": (128, 256, 96, 12), "hs256b": (256, 256, 96, 12)}.items():
        m = rg.HS_layer(Cin, Co, support_num=S).eval()
        for i, (kk, v) in enumerate(m.state_dict().items()):
            oc.hash_fill_(v, 7700 + 10 * Cin // 128 + 100 * Co // 128 + i, 0.05)
        X = torch.relu(oc.hash_tensor((2, n, Cin), 7800 + Cin + Co, 1.0))
        with torch.no_grad():
            out[tag] = m(xyz[:, :n].contiguous(), X, kk_).numpy()
    np.savez_compressed(os.path.join(GOLD, "exact_layers.npz"), **out)
    print("exact_layers:", {k_: v.shape for k_, v in out.items()})


def stack(debug_dump=None):
    FLAGS.train = 0
    torch.manual_seed(0)                       # the reference's own initialisation (as stack_refinit_eval_1028)
    from network.fs_net_repo.PoseNet9D import PoseNet9D as RefPoseNet9D
    net = RefPoseNet9D().eval()
    fr = net.face_recon
    B, seed = 2, 81
    pts = oc.hash_tensor((B, 1028, 3), seed, 0.05)
    pts[:, :, 2] += 0.8
    obj = torch.from_numpy((oc.hash_unit(B, seed + 1) * 6).astype(np.int64)).float().view(B, 1)
    grabbed = {}
    hooks = []
    for nm in ("conv_0", "conv_1", "conv_2", "conv_3", "conv_4", "bn1", "bn2", "bn3"):
        hooks.append(getattr(fr, nm).register_forward_hook(lambda mod, i, o, nm=nm: grabbed.__setitem__(nm, o.detach().clone())))
    torch.manual_seed(1)
    with torch.no_grad():
        _, _, feat = fr(pts - pts.mean(dim=1, keepdim=True), obj)                  # PoseNet9D.py:25 centres the cloud
    for h_ in hooks:
        h_.remove()
    out = {"meta": np.array([B, 1028, seed, 0], np.int64), "feat": feat.reshape(-1)[::211].numpy().copy()}
    for nm, t in grabbed.items():
        out[nm] = t.reshape(-1)[::53].numpy().copy()
    np.savez_compressed(os.path.join(GOLD, "exact_stack_1028.npz"), **out)
    print("exact_stack_1028:", {k_: v.shape for k_, v in out.items()})
    if debug_dump:
        os.makedirs(os.path.dirname(debug_dump), exist_ok=True)
        np.savez(debug_dump, feat=feat.numpy(), **{k_: v.numpy() for k_, v in grabbed.items()})


if __name__ == "__main__":
    statements()
    topk_ties()
    layers()
    dd = sys.argv[sys.argv.index("--debug-dump") + 1] if "--debug-dump" in sys.argv else None
    stack(dd)
