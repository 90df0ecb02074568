"""Diagnostic (not a test; build container or any CPU): how far the CPU ORACLE's own free-running pose / size outputs move
when its input cloud is perturbed by 1 ulp (relative 1.2e-7), at the golden full-stack cases.  This is the conditioning of
the reference algorithm itself -- neighbour selection is discontinuous -- and the yardstick for the free-running GPU
test's bound (tests/test_gpu_stack.py::test_posenet9d_free_running_1028).

    python tools/oracle_free_running_sensitivity.py stack_eval_1028 stack_evalflags_trainbn_1028
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT]
import numpy as np
import torch

import ref_cpu as oc
from conftest import golden
from hs_pose_amd.config import FLAGS

OUTS = ["p_green_R", "p_red_R", "f_green_R", "f_red_R", "Pred_T", "Pred_s"]
torch.set_num_threads(8)


def run(name, noise_seed):
    g = golden(name)
    train_flag, B, N, seed, bn_training = (int(v) for v in g["meta"])
    FLAGS.reset(); FLAGS.train = train_flag
    from hs_pose_amd.PoseNet9D import PoseNet9D
    sd = PoseNet9D().state_dict()
    oc.fill_state_closed_form(sd)
    pts = oc.hash_tensor((B, N, 3), seed, 0.05)
    pts[:, :, 2] += 0.8
    if noise_seed is not None:
        pts = pts * (1 + oc.hash_tensor(tuple(pts.shape), noise_seed, 1.2e-7))
    obj = torch.from_numpy((oc.hash_unit(B, seed + 1) * 6).astype(np.int64)).float().view(B, 1)
    pidx = [torch.from_numpy(g["pool_idx0"].astype(np.int64)), torch.from_numpy(g["pool_idx1"].astype(np.int64))]
    lists, real = [], oc.knn_index

    def rec(x, k):
        idx = real(x, k)
        if x.shape[-1] != 3:
            lists.append(idx)
        return idx
    oc.knn_index = rec
    try:
        with torch.no_grad():
            o = oc.posenet9d(sd, pts, obj, pidx, train_heads=bool(train_flag), bn_training=bool(bn_training))
    finally:
        oc.knn_index = real
    return {k: o[k] for k in OUTS}, lists, g


for name in sys.argv[1:]:
    base, l0, g = run(name, None)
    gold = max(float((base[k] - torch.from_numpy(g["out." + k])).abs().max()) for k in OUTS)
    print(f"{name}: oracle (this run, 8 threads) vs golden fixture: {gold:.2e}")
    for ns in (999, 1000, 1001):
        o, l1, _ = run(name, ns)
        agree = [round(float((a == b).all(dim=2).float().mean()), 4) for a, b in zip(l0, l1)]
        errs = {k: float(f"{float((o[k] - base[k]).abs().max()):.2e}") for k in OUTS}
        print(f"  1-ulp input noise (seed {ns}): rows with identical neighbour sets per HS layer {agree}; max abs output change {errs}")
