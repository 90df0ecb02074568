"""
oracle/gen_golden.py -- TEST INFRASTRUCTURE ONLY.  Runs ONLY in the build container.

Imports the reference implementation from /root/reference (with the tiny import stubs under
oracle/stubs/ for packages the image lacks), and
  1. checks oracle/ref_cpu.py (torch restatement) and oracle/hsp_oracle.c (index oracle) against it,
  2. writes small golden vectors (inputs are closed-form hash fills, so mostly only OUTPUTS are
     stored) to tests/golden/*.npz plus tests/golden/manifest.json.
The reference's source never enters this repo: fixtures hold numbers only.

usage:  python oracle/gen_golden.py            (re-generates every fixture; ~1-2 min on 8 cores)
"""
import ctypes
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path[:0] = [os.path.join(HERE, "stubs"), REF, HERE]

import numpy as np
import torch

import config.config  # noqa: F401  (reference flag definitions)
from absl import flags

FLAGS = flags.FLAGS
import network.fs_net_repo.gcn3d as rg  # reference hot path
from network.fs_net_repo.FaceRecon import FaceRecon as RefFaceRecon
from network.fs_net_repo.PoseNet9D import PoseNet9D as RefPoseNet9D

import ref_cpu as oc

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
# ONE thread: MKL/OpenMP reductions are then bit-reproducible run to run, so regenerating writes byte-identical
# fixtures (gradients included) and the oracle-vs-reference gradient cross-checks below cannot flake.
torch.set_num_threads(int(os.environ.get("HSP_GOLDEN_THREADS", "1")))
manifest = {"torch": torch.__version__, "threads": torch.get_num_threads(), "files": {}}


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


subprocess.check_call(["make", "-C", HERE, "-s"])
clib = ctypes.CDLL(os.path.join(HERE, "libhsp_oracle.so"))


def c_knn(x, k, drop_first=1):
    xn = np.ascontiguousarray(x.numpy(), dtype=np.float32)
    B, N, C = xn.shape
    out = np.empty((B, N, k), np.int32)
    rc = clib.hsp_oracle_knn(P(xn), B, N, C, k, drop_first, P(out), None)
    assert rc == 0
    return out


def c_nn1(t, s):
    tn = np.ascontiguousarray(t.numpy(), dtype=np.float32)
    sn = np.ascontiguousarray(s.numpy(), dtype=np.float32)
    out = np.empty((tn.shape[0], tn.shape[1]), np.int32)
    clib.hsp_oracle_nn1(P(tn), tn.shape[1], P(sn), sn.shape[1], tn.shape[0], tn.shape[2], P(out))
    return out


def save(name, **arrs):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrs)
    manifest["files"][name] = {k: [list(v.shape), str(v.dtype)] for k, v in arrs.items()}
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def tie_free(x, k=20):
    """no exact ties among the k+2 smallest distances of any row (the selection and its boundary)."""
    d = oc.knn_dist(x).numpy()
    s = np.sort(d, axis=-1)[..., :k + 2]
    return int((np.diff(s, axis=-1) == 0).sum()) == 0


# ------------------------------------------------------------------------------------------------
# a-1 / a-2: neighbour indices
# ------------------------------------------------------------------------------------------------
print("[knn]")
torch.manual_seed(0)
v_cfg1 = torch.randn(1, 256, 3)                   # BASELINE config 1 (SURVEY 8c known answer)
ref_idx = rg.get_neighbor_index(v_cfg1, 20)
assert ref_idx[0, 0, :5].tolist() == [121, 239, 166, 46, 230]
assert torch.equal(ref_idx, oc.knn_index(v_cfg1, 20))
assert tie_free(v_cfg1)
assert np.array_equal(c_knn(v_cfg1, 20), ref_idx.numpy())
save("knn_cfg1", x=v_cfg1.numpy(), idx=ref_idx.numpy().astype(np.int16))

knn_cases = {
    # name: (shape, seed, scale, offset, k)
    "knn_xyz_1028": ((2, 1028, 3), 11, 0.1, 0.0, 20),
    "knn_xyz_257": ((3, 257, 3), 12, 0.1, 0.0, 20),
    "knn_xyz_64_k8": ((4, 64, 3), 13, 0.1, 0.0, 8),
    "knn_xyz_1028_k4": ((2, 1028, 3), 11, 0.1, 0.0, 4),
    "knn_xyz_offset": ((2, 300, 3), 14, 0.05, 0.8, 20),   # uncentred cloud: cancellation-heavy
    "knn_feat128_1028": ((1, 1028, 128), 21, 1.0, 0.0, 20),
    "knn_feat128_257": ((2, 257, 128), 22, 1.0, 0.0, 20),
    "knn_feat256_257": ((2, 257, 256), 23, 1.0, 0.0, 20),
    "knn_feat256_64_k8": ((3, 64, 256), 24, 1.0, 0.0, 8),
    "knn_feat16_128_k8": ((2, 128, 16), 25, 1.0, 0.0, 8),
    "knn_feat32_16_k2": ((2, 16, 32), 26, 1.0, 0.0, 2),
    "knn_relu_feat128_257": ((2, 257, 128), 27, 1.0, 0.0, 20),  # post-ReLU-like (half zeros)
}
for name, (shape, seed, scale, off, k) in knn_cases.items():
    x = oc.hash_tensor(shape, seed, scale, off)
    if "relu" in name:
        x = torch.relu(x)
    r = rg.get_neighbor_index(x, k)
    assert torch.equal(r, oc.knn_index(x, k)), name
    tf = tie_free(x, k)
    ci = c_knn(x, k)
    eq = np.array_equal(ci, r.numpy())
    print(f"  {name}: tie_free={tf} C-oracle==reference: {eq}")
    if tf:
        assert eq, name
    else:
        # exact ties present (torch.topk order unspecified): the two selections must still pick the
        # same distance VALUES rank by rank -- only equal-distance candidates may be swapped
        d = oc.knn_dist(x)
        dv_ref = torch.gather(d, 2, r).numpy()
        dv_c = torch.gather(d, 2, torch.from_numpy(ci).long()).numpy()
        assert np.array_equal(dv_ref, dv_c), name
        assert "offset" in name
    save(name, idx=r.numpy().astype(np.int16), meta=np.array([*shape, seed, k, int(tf)], np.int64),
         scale_off=np.array([scale, off], np.float64))

# prefix property the build relies on: top-4 neighbours == first 4 of the top-20 (tie-free input)
x = oc.hash_tensor((2, 1028, 3), 11, 0.1)
assert torch.equal(rg.get_neighbor_index(x, 4), rg.get_neighbor_index(x, 20)[:, :, :4])

print("[nearest]")
tgt = oc.hash_tensor((2, 1028, 3), 31, 0.1)
perm = torch.from_numpy(np.argsort(oc.hash_unit(1028, 32)))
for name, m in (("nn1_1028_257", 257), ("nn1_1028_64", 64)):
    src = tgt[:, perm[:m], :].contiguous()
    r = rg.get_nearest_index(tgt, src)
    assert torch.equal(r, oc.nearest_index(tgt, src))
    assert np.array_equal(c_nn1(tgt, src), r.squeeze(-1).numpy()), name
    save(name, idx=r.squeeze(-1).numpy().astype(np.int16), perm=perm[:m].numpy().astype(np.int16))

# ------------------------------------------------------------------------------------------------
# a-4 / a-6 / a-7: layers, reduced width with full outputs + gradients, and full width sampled
# ------------------------------------------------------------------------------------------------
print("[layers]")


def state_of(mod):
    return {k: v for k, v in mod.state_dict().items()}


def fill_module(mod):
    sd = mod.state_dict()
    oc.fill_state_closed_form(sd)     # in place: state_dict tensors alias the parameters
    return sd


def gclose(a, b):
    """oracle-vs-reference gradient check: same math, but threaded MKL reductions are not bit-stable
    run to run, so compare to 1e-5 of the tensor's scale."""
    scale = max(b.abs().max().item(), 1e-12)
    return (a - b).abs().max().item() <= 1e-5 * scale


def grads_of(mod):
    return {k: p.grad.detach().clone() for k, p in mod.named_parameters() if p.grad is not None}


def run_surface(K, S, N, k, B, seed, name, full):
    m = rg.HSlayer_surface(kernel_num=K, support_num=S)
    sd = fill_module(m)
    xyz = oc.hash_tensor((B, N, 3), seed, 0.1)
    up = oc.hash_tensor((B, N, K), seed + 1, 1.0)
    out = m(xyz, k)
    (out * up).sum().backward()
    g = grads_of(m)
    # oracle check (forward exact op sequence; gradients through autograd of the restatement)
    p = {k_: v.detach().clone().requires_grad_(True) for k_, v in sd.items()}
    o2 = oc.surface_layer(p, "", xyz, k, S)
    assert torch.equal(o2, out), name
    (o2 * up).sum().backward()
    for k_ in g:
        assert gclose(p[k_].grad, g[k_]), (name, k_, (p[k_].grad - g[k_]).abs().max().item())
    arrs = {"meta": np.array([K, S, N, k, B, seed], np.int64)}
    if full:
        arrs["out"] = out.detach().numpy()
    else:
        arrs["out_sample"] = out.detach().reshape(-1)[::997].numpy().copy()
        arrs["out_sum"] = np.array([out.double().sum().item(), out.double().abs().sum().item()])
        arrs["out_chmean"] = out.detach().mean(dim=(0, 1)).numpy()
    for k_, v in g.items():
        arrs["grad." + k_] = v.numpy()
    save(name, **arrs)


def run_hs(Cin, Cout, S, N, k, B, seed, name, full):
    m = rg.HS_layer(Cin, Cout, support_num=S)
    sd = fill_module(m)
    while not tie_free(torch.relu(oc.hash_tensor((B, N, Cin), seed + 2, 1.0)), k):
        seed += 100      # exact fp32 ties among the k+2 nearest: pick the next input seed
    xyz = oc.hash_tensor((B, N, 3), seed, 0.1)
    fmap = torch.relu(oc.hash_tensor((B, N, Cin), seed + 2, 1.0)).requires_grad_(True)
    up = oc.hash_tensor((B, N, Cout), seed + 1, 1.0)
    out = m(xyz, fmap, k)
    (out * up).sum().backward()
    g = grads_of(m)
    gx = fmap.grad.detach().clone()
    p = {k_: v.detach().clone().requires_grad_(True) for k_, v in sd.items()}
    f2 = fmap.detach().clone().requires_grad_(True)
    o2, idx = oc.hs_layer(p, "", xyz, f2, k, S, return_idx=True)
    assert torch.equal(o2, out), name
    (o2 * up).sum().backward()
    for k_ in g:
        assert gclose(p[k_].grad, g[k_]), (name, k_, (p[k_].grad - g[k_]).abs().max().item())
    assert gclose(f2.grad, gx)
    assert tie_free(fmap.detach(), k), name
    arrs = {"meta": np.array([Cin, Cout, S, N, k, B, seed], np.int64), "knn_idx": idx.numpy().astype(np.int16)}
    if full:
        arrs["out"] = out.detach().numpy()
        arrs["grad_fmap"] = gx.numpy()
        for k_, v in g.items():
            arrs["grad." + k_] = v.numpy()
    else:
        arrs["out_sample"] = out.detach().reshape(-1)[::997].numpy().copy()
        arrs["out_sum"] = np.array([out.double().sum().item(), out.double().abs().sum().item()])
        arrs["out_chmean"] = out.detach().mean(dim=(0, 1)).numpy()
        arrs["grad_fmap_sample"] = gx.reshape(-1)[::997].numpy().copy()
        for k_, v in g.items():
            arrs["gradnorm." + k_] = np.array([v.double().norm().item(), v.double().sum().item()])
            arrs["gradsample." + k_] = v.reshape(-1)[::499].numpy().copy()
    save(name, **arrs)


run_surface(16, 3, 128, 8, 2, 41, "surface_small", True)
run_surface(128, 7, 257, 20, 1, 42, "surface_full", False)
run_hs(16, 32, 3, 128, 8, 2, 51, "hs_small", True)
run_hs(128, 128, 7, 257, 20, 1, 52, "hs_full_128", False)
run_hs(256, 512, 7, 64, 8, 2, 53, "hs_full_512", False)

# a-9: pool with a given sample_idx, and the seed -> randperm pair
print("[pool]")
torch.manual_seed(1)
perm_a = torch.randperm(1028)[:257]
perm_b = torch.randperm(257)[:64]
torch.manual_seed(1)
drawn = oc.draw_pool_indices(1028)
assert torch.equal(drawn[0], perm_a) and torch.equal(drawn[1], perm_b)
xyz = oc.hash_tensor((2, 1028, 3), 61, 0.1)
fmap = oc.hash_tensor((2, 1028, 32), 62, 1.0)
pool = rg.Pool_layer(4, 4)
torch.manual_seed(1)
vp, fp = pool(xyz, fmap)
vp2, fp2 = oc.pool_layer(xyz, fmap, perm_a)
assert torch.equal(vp, vp2) and torch.equal(fp, fp2)
save("pool_1028", perm_seed1_a=perm_a.numpy().astype(np.int16), perm_seed1_b=perm_b.numpy().astype(np.int16),
     v_pool=vp.numpy(), f_pool=fp.numpy())

# ------------------------------------------------------------------------------------------------
# a-10 .. a-12: FaceRecon / PoseNet9D
# ------------------------------------------------------------------------------------------------
print("[stack]")


class KnnRecorder:
    """records the reference's internal get_neighbor_index results (feature-space calls only) so the
    GPU stack can be teacher-forced with the very same neighbour sets ("selection
    discontinuity")."""

    def __init__(self):
        self.feat_idx = []
        self._orig = rg.get_neighbor_index

    def __enter__(self):
        def rec(vertices, neighbor_num):
            out = self._orig(vertices, neighbor_num)
            if vertices.shape[-1] != 3:
                self.feat_idx.append(out.clone())
            return out
        rg.get_neighbor_index = rec
        return self

    def __exit__(self, *a):
        rg.get_neighbor_index = self._orig


def run_stack(train_flag, B, N, seed, name, bn_training, with_grads):
    FLAGS.train = train_flag
    net = RefPoseNet9D()
    sd = fill_module(net)
    keys = {k_: list(v.shape) for k_, v in sd.items()}
    net.train(bn_training)
    for mod in net.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    pts = oc.hash_tensor((B, N, 3), seed, 0.05)
    pts[:, :, 2] += 0.8
    obj = torch.from_numpy((oc.hash_unit(B, seed + 1) * 6).astype(np.int64)).float().view(B, 1)
    centred = pts - pts.mean(dim=1, keepdim=True)
    torch.manual_seed(1)
    with KnnRecorder() as rec:
        outs = net(pts, obj)
    names = ["recon", "face_normal", "face_dis", "face_f", "p_green_R", "p_red_R", "f_green_R", "f_red_R", "Pred_T", "Pred_s"]
    outs = dict(zip(names, outs))
    # oracle restatement on a copy of the (pre-step) state
    sd0 = fill_module(RefPoseNet9D())
    p = {k_: v.detach().clone() for k_, v in sd0.items()}
    for k_ in p:
        if p[k_].is_floating_point() and "running" not in k_:
            p[k_].requires_grad_(True)
    torch.manual_seed(1)
    pidx = oc.draw_pool_indices(N)
    o = oc.posenet9d(p, pts, obj, pidx, train_heads=bool(train_flag), bn_training=bn_training)
    for n_ in names[4:]:
        assert torch.equal(o[n_], outs[n_]), (name, n_)
    arrs = {"meta": np.array([train_flag, B, N, seed, int(bn_training)], np.int64),
            "centred": centred.numpy(),        # bit-exact stack input (the GPU mean may round differently)
            "pool_idx0": pidx[0].numpy().astype(np.int16), "pool_idx1": pidx[1].numpy().astype(np.int16)}
    assert len(rec.feat_idx) == 4
    for li, fi in enumerate(rec.feat_idx):
        arrs[f"featknn{li + 1}"] = fi.numpy().astype(np.int16)
    for n_ in names[4:]:
        arrs["out." + n_] = outs[n_].detach().numpy()
    if train_flag:
        for n_ in names[:4]:
            assert torch.equal(o[n_], outs[n_]), (name, n_)
            arrs["outsample." + n_] = outs[n_].detach().reshape(-1)[::211].numpy().copy()
    feat = o["feat"].detach()
    arrs["feat_sample"] = feat.reshape(-1)[::1009].numpy().copy()
    arrs["feat_sum"] = np.array([feat.double().sum().item(), feat.double().abs().sum().item()])
    arrs["feat_chmean"] = feat.mean(dim=(0, 1)).numpy()
    if with_grads:
        # backward of the HS stack alone from a closed-form dfeat (unit U1 of SURVEY 8d)
        net2 = RefPoseNet9D(); fill_module(net2); net2.train(bn_training)
        torch.manual_seed(1)
        _, _, f2 = net2.face_recon(centred, obj)
        dfeat = oc.hash_tensor(tuple(f2.shape), seed + 5, 1.0)
        (f2 * dfeat).sum().backward()
        for k_, prm in net2.face_recon.named_parameters():
            if prm.grad is not None:
                arrs["gradnorm." + k_] = np.array([prm.grad.double().norm().item(), prm.grad.double().sum().item()])
                arrs["gradsample." + k_] = prm.grad.reshape(-1)[::499].numpy().copy()
        if bn_training:
            for k_, buf in net2.face_recon.named_buffers():
                if "running" in k_ and k_.startswith("bn"):
                    arrs["bnstat." + k_] = buf.numpy().copy()
        # oracle gradient check
        (o["feat"] * dfeat).sum().backward()
        for k_, prm in net2.face_recon.named_parameters():
            if prm.grad is not None:
                assert gclose(p["face_recon." + k_].grad, prm.grad), (name, k_, (p["face_recon." + k_].grad - prm.grad).abs().max().item(), prm.grad.abs().max().item())
    save(name, **arrs)
    return keys


keys_eval = run_stack(0, 2, 256, 71, "stack_eval_256", False, False)
run_stack(0, 2, 1028, 74, "stack_eval_1028", False, False)
run_stack(0, 4, 1028, 72, "stack_evalflags_trainbn_1028", True, True)     # B >= 4: train-mode BN on (B,256) head rows
keys_train = run_stack(1, 4, 256, 73, "stack_train_256", True, True)
with open(os.path.join(GOLD, "state_keys.json"), "w") as f:
    json.dump({"train": keys_train, "eval": keys_eval}, f, indent=0, sort_keys=True)
assert len(keys_train) == 160 and len(keys_eval) == 107, (len(keys_train), len(keys_eval))
FLAGS.train = 1

# a-15 (FPS) and a-16 (Chamfer) are pinned by oracle/gen_golden_chamfer_fps.py: the numpy helper in both dtypes and the
# reference's own chamfer_distance.cpp compiled as is (oracle/_ref/cd_ref.so).

with open(os.path.join(GOLD, "manifest.json"), "w") as f:
    json.dump(manifest, f, indent=1, sort_keys=True)
tot = sum(os.path.getsize(os.path.join(GOLD, f_)) for f_ in os.listdir(GOLD))
print(f"golden total: {tot / 1024:.1f} KiB")
