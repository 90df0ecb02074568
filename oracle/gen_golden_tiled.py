"""
oracle/gen_golden_tiled.py -- TEST INFRASTRUCTURE ONLY.  Runs ONLY in the build container.

TILED clouds through the reference's stack.  A real crop with fewer than 1028 valid pixels is tiled up to 1028 points by the
reference's loader (datasets/load_data.py:306-318: np.tile + the leading remainder), so exact xyz duplicates -- and with them
exactly equal distances everywhere in get_neighbor_index (network/fs_net_repo/gcn3d.py:15-24) -- are the NORMAL case on real
data.  What torch.topk returns among equal distances is decided by libstdc++'s nth_element / partial_sort (ATen TopKImpl.h), and
it differs between the k + 1 = 21 search of the layers and the k + 1 = 5 search of Pool_layer (gcn3d.py:236): the k = 4 list is
NOT the prefix of the k = 20 list on such a cloud.

Imports the reference from /root/reference (stubs under oracle/stubs/), one torch thread, reference-initialised weights
(torch.manual_seed(0), as stack_refinit_*), and writes

  tests/golden/exact_stack_tiled_1028.npz   eval mode, B = 2: a 400-point and a 1000-point cloud tiled to 1028.  Every xyz
        neighbour list the forward asks for (k = 20 and k = 4 at N0 = 1028 and N1 = 257, k = 8 at N2 = 64), the four feature-space
        lists, both Pool_layer outputs, strided samples of conv_0 ... conv_4 / feat, the six pose / size outputs.
  tests/golden/stack_tiled_trainbn_1028.npz  train-mode BatchNorm, B = 4 (400 / 1000 / 257 / 600 base points): the same lists,
        the six outputs, and the backward of the HS stack from a closed-form dfeat (gradient samples + norms, BatchNorm running
        statistics) -- with exact duplicates among a point's neighbours the max over the neighbours (gcn3d.py:178) ties, and
        torch.max's first-index rule decides where the gradient goes.

The clouds are closed-form (splitmix64 fills, oracle/ref_cpu.py::hash_tensor), so the fixtures hold outputs only.  The
reference's source never enters this repo.

usage:  python oracle/gen_golden_tiled.py
"""
import os
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
import network.fs_net_repo.gcn3d as rg
from network.fs_net_repo.PoseNet9D import PoseNet9D as RefPoseNet9D

import ref_cpu as oc

GOLD = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(1)
NAMES = ["recon", "face_normal", "face_dis", "face_f", "p_green_R", "p_red_R", "f_green_R", "f_red_R", "Pred_T", "Pred_s"]
N_PTS = 1028


tiled_batch = oc.tiled_batch


class Recorder:
    """every get_neighbor_index result of a forward: xyz lists keyed by (N, k) (the same search is repeated by the RF-P branch
    and the ORL branches: asserted identical), feature-space lists in call order"""

    def __init__(self):
        self.xyz, self.feat = {}, []
        self._orig = rg.get_neighbor_index

    def __enter__(self):
        def rec(vertices, neighbor_num):
            out = self._orig(vertices, neighbor_num)
            if vertices.shape[-1] == 3:
                key = (vertices.shape[1], neighbor_num)
                if key in self.xyz:
                    assert torch.equal(self.xyz[key], out), key
                self.xyz[key] = out.clone()
            else:
                self.feat.append(out.clone())
            return out
        rg.get_neighbor_index = rec
        return self

    def __exit__(self, *a):
        rg.get_neighbor_index = self._orig


def build(bn_training):
    FLAGS.train = 0
    torch.manual_seed(0)
    net = RefPoseNet9D()
    net.train(bn_training)
    for mod in net.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return net


def hook_stack(fr, grabbed):
    hooks = []
    for nm in ("conv_0", "conv_1", "conv_2", "conv_3", "conv_4", "bn1", "bn2", "bn3"):
        hooks.append(getattr(fr, nm).register_forward_hook(lambda mod, i, o, nm=nm: grabbed.__setitem__(nm, o.detach().clone())))
    for nm in ("pool_1", "pool_2"):
        hooks.append(getattr(fr, nm).register_forward_hook(
            lambda mod, i, o, nm=nm: grabbed.__setitem__(nm, (o[0].detach().clone(), o[1].detach().clone()))))
    return hooks


def tie_report(pts, lists):
    """how tie-ridden the case is: rows of the N0 search holding an exact tie among their 22 nearest, and rows where the k = 4
    list is not the prefix of the k = 20 list / not the same SET as its first four"""
    l20, l4 = lists[(N_PTS, 20)], lists[(N_PTS, 4)]
    pre = l20[:, :, :4]
    differ = (pre != l4).any(-1).float().mean().item()
    differ_set = (torch.sort(pre, -1)[0] != torch.sort(l4, -1)[0]).any(-1).float().mean().item()
    return differ, differ_set


def common_arrays(grabbed, rec, outs, feat):
    arrs = {}
    for (n, k), v in rec.xyz.items():
        arrs[f"xyz_n{n}_k{k}"] = v.numpy().astype(np.int16)
    assert len(rec.feat) == 4
    for li, fi in enumerate(rec.feat):
        arrs[f"featknn{li + 1}"] = fi.numpy().astype(np.int16)
    for nm in ("conv_0", "conv_1", "conv_2", "conv_3", "conv_4", "bn1", "bn2", "bn3"):
        arrs[nm] = grabbed[nm].reshape(-1)[::53].numpy().copy()
    for nm in ("pool_1", "pool_2"):
        arrs[nm + ".vertices"] = grabbed[nm][0].numpy().copy()
        arrs[nm + ".feature"] = grabbed[nm][1].reshape(-1)[::29].numpy().copy()
    arrs["feat"] = feat.reshape(-1)[::211].numpy().copy()
    for n_ in NAMES[4:]:
        arrs["out." + n_] = outs[n_].detach().numpy()
    return arrs


def eval_case(name, bases, seed):
    net = build(False)
    fr = net.face_recon
    B = len(bases)
    pts = tiled_batch(bases, seed)
    obj = torch.from_numpy((oc.hash_unit(B, seed + 1) * 6).astype(np.int64)).float().view(B, 1)
    grabbed = {}
    hooks = hook_stack(fr, grabbed)
    feats = []
    hooks.append(fr.register_forward_hook(lambda mod, i, o: feats.append(o[2].detach().clone())))
    torch.manual_seed(1)                                   # Pool_layer randperm stream (gcn3d.py:243)
    with Recorder() as rec, torch.no_grad():
        outs = dict(zip(NAMES, net(pts, obj)))
    for h_ in hooks:
        h_.remove()
    arrs = {"meta": np.array([B, N_PTS, seed, 0] + list(bases), np.int64)}
    arrs.update(common_arrays(grabbed, rec, outs, feats[0]))
    d, ds = tie_report(pts, rec.xyz)
    arrs["k4_vs_k20_prefix"] = np.array([d, ds])
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **arrs)
    print(f"{name}: {os.path.getsize(os.path.join(GOLD, name + '.npz')) / 1024:.1f} KiB; xyz lists {sorted(rec.xyz)}; rows whose "
          f"k = 4 list differs from the k = 20 prefix {d:.3f} (as a set {ds:.3f})")


def train_case(name, bases, seed):
    B = len(bases)
    pts = tiled_batch(bases, seed)
    obj = torch.from_numpy((oc.hash_unit(B, seed + 1) * 6).astype(np.int64)).float().view(B, 1)
    net = build(True)
    fr = net.face_recon
    grabbed = {}
    hooks = hook_stack(fr, grabbed)
    feats = []
    hooks.append(fr.register_forward_hook(lambda mod, i, o: feats.append(o[2].detach().clone())))
    torch.manual_seed(1)
    with Recorder() as rec, torch.no_grad():
        outs = dict(zip(NAMES, net(pts, obj)))
    for h_ in hooks:
        h_.remove()
    arrs = {"meta": np.array([B, N_PTS, seed, 1] + list(bases), np.int64)}
    arrs.update(common_arrays(grabbed, rec, outs, feats[0]))
    d, ds = tie_report(pts, rec.xyz)
    arrs["k4_vs_k20_prefix"] = np.array([d, ds])
    # unit U1 backward on a fresh copy (train-mode BatchNorm moved the running statistics above): feat from the centred cloud,
    # closed-form dfeat, gradients of every HS-stack parameter
    net2 = build(True)
    centred = pts - pts.mean(dim=1, keepdim=True)          # PoseNet9D.py:25
    arrs["centred"] = centred.numpy()
    torch.manual_seed(1)
    with Recorder() as rec2:
        _, _, f2 = net2.face_recon(centred, obj)
    for key, v in rec.xyz.items():
        assert torch.equal(v, rec2.xyz[key]), key
    for a, b in zip(rec.feat, rec2.feat):
        assert torch.equal(a, b)
    dfeat = oc.hash_tensor(tuple(f2.shape), seed + 5, 1.0)
    (f2 * dfeat).sum().backward()
    for k_, prm in net2.face_recon.named_parameters():
        if prm.grad is not None:
            arrs["gradnorm." + k_] = np.array([prm.grad.double().norm().item(), prm.grad.double().sum().item()])
            arrs["gradsample." + k_] = prm.grad.reshape(-1)[::499].numpy().copy()
    for k_, buf in net2.face_recon.named_buffers():
        if "running" in k_ and k_.startswith("bn"):
            arrs["bnstat." + k_] = buf.numpy().copy()
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **arrs)
    print(f"{name}: {os.path.getsize(os.path.join(GOLD, name + '.npz')) / 1024:.1f} KiB; rows whose k = 4 list differs from the "
          f"k = 20 prefix {d:.3f} (as a set {ds:.3f})")


if __name__ == "__main__":
    eval_case("exact_stack_tiled_1028", (400, 1000), 91)
    train_case("stack_tiled_trainbn_1028", (400, 1000, 257, 600), 93)
