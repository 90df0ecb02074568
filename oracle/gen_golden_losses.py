"""
oracle/gen_golden_losses.py -- TEST INFRASTRUCTURE ONLY.  Runs ONLY in the build container.

Golden vectors for the training losses and the on-device augmentation: the reference's four loss
modules (losses/*.py) wired exactly as network/HSPose.py:84-181 wires them, run on the closed-form batch of
ref_cpu.loss_case (one object per symmetry class of the dataset), and HSPose.data_augment (HSPose.py:185-256) run
under a seeded CPU generator on ref_cpu.augment_case.  Outputs: the 19 weighted loss terms, the gradient of their sum
with respect to every network output, the augmented clouds / poses -> tests/golden/losses_*.npz.

usage:  python oracle/gen_golden_losses.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path[:0] = [os.path.join(HERE, "stubs"), REF, HERE]

import numpy as np
import torch

import config.config  # noqa: F401
from absl import flags

FLAGS = flags.FLAGS
import network.HSPose as RH
from engine.organize_loss import control_loss
from losses.fs_net_loss import fs_net_loss
from losses.geometry_loss import geo_transform_loss
from losses.prop_loss import prop_rot_loss
from losses.recon_loss import recon_6face_loss
from tools.training_utils import get_gt_v

import ref_cpu as oc

GOLD = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(1)


def save(name, **arrs):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrs)
    mpath = os.path.join(GOLD, "manifest.json")
    man = json.load(open(mpath))
    man["files"][name] = {k: [list(v.shape), str(v.dtype)] for k, v in arrs.items()}
    json.dump(man, open(mpath, "w"), indent=1, sort_keys=True)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def reference_losses(gt, pred):
    """the body of network/HSPose.py:84-181 with the reference modules"""
    fs, rc, ge, pr = control_loss('PoseNet_only')
    sym, obj_id, PC = gt["sym"], gt["obj_id"], gt["PC"]
    gt_green_v, gt_red_v = get_gt_v(gt["gt_R"])
    p = pred
    fsnet = fs_net_loss()(fs, {'Rot1': p["p_green_R"], 'Rot1_f': p["f_green_R"], 'Rot2': p["p_red_R"], 'Rot2_f': p["f_red_R"],
                               'Recon': p["recon"], 'Tran': p["Pred_T"], 'Size': p["Pred_s"]},
                          {'Rot1': gt_green_v, 'Rot2': gt_red_v, 'Recon': PC, 'Tran': gt["gt_t"], 'Size': gt["gt_s"]}, sym)
    prop = prop_rot_loss()(pr, {'Recon': p["recon"], 'Rot1': p["p_green_R"], 'Rot2': p["p_red_R"], 'Tran': p["Pred_T"],
                                'Scale': p["Pred_s"], 'Rot1_f': p["f_green_R"].detach(), 'Rot2_f': p["f_red_R"].detach()},
                           {'Points': PC, 'R': gt["gt_R"], 'T': gt["gt_t"], 'Mean_shape': gt["mean_shape"]}, sym)
    recon = recon_6face_loss()(rc, {'F_n': p["face_normal"], 'F_d': p["face_dis"], 'F_c': p["face_f"], 'Rot1': p["p_green_R"],
                                    'Rot1_f': p["f_green_R"].detach(), 'Rot2': p["p_red_R"], 'Rot2_f': p["f_red_R"].detach(),
                                    'Tran': p["Pred_T"], 'Size': p["Pred_s"]},
                               {'R': gt["gt_R"], 'T': gt["gt_t"], 'Size': gt["gt_s"], 'Mean_shape': gt["mean_shape"], 'Points': PC},
                               sym, obj_id)
    geo = geo_transform_loss()(ge, {'Rot1': p["p_green_R"], 'Rot2': p["p_red_R"], 'Tran': p["Pred_T"], 'Size': p["Pred_s"],
                                    'Rot1_f': p["f_green_R"].detach(), 'Rot2_f': p["f_red_R"].detach()},
                               {'Points': PC, 'R': gt["gt_R"], 'T': gt["gt_t"], 'Mean_shape': gt["mean_shape"]}, sym)
    return {'fsnet_loss': fsnet, 'recon_loss': recon, 'geo_loss': geo, 'prop_loss': prop}


print("losses")
for name, fl in (("losses_l1", "l1"), ("losses_smoothl1", "smoothl1")):
    FLAGS.fsnet_loss_type = fl
    gt, pred = oc.loss_case()
    ld = reference_losses(gt, pred)
    total = sum(ld['fsnet_loss'].values()) + sum(ld['recon_loss'].values()) + sum(ld['geo_loss'].values()) + sum(ld['prop_loss'].values())
    total.backward()
    out = {}
    n = 0
    for grp, d in ld.items():
        for k, v in d.items():
            out[f"{grp}.{k}"] = np.asarray(v.detach().numpy(), np.float32).reshape(-1)
            n += 1
    print(f"  {name}: {n} terms, total {float(total):.6f}")
    assert n == 19
    for k, v in pred.items():
        out["grad." + k] = v.grad.numpy()
    out["total"] = np.array([float(total)], np.float32)
    save(name, **out)
FLAGS.fsnet_loss_type = "l1"

print("get_gt_v / augmentation")
gt = oc.augment_case()
g, r = get_gt_v(gt["gt_R"])
aug = {}
for tag, pro in (("half", 0.6), ("all", 1.1), ("none", -1.0)):
    FLAGS.aug_bb_pro = FLAGS.aug_rt_pro = FLAGS.aug_bc_pro = FLAGS.aug_pc_pro = pro
    torch.manual_seed(5)
    PC, R, t, s = RH.HSPose.data_augment(None, gt["PC"].clone(), gt["gt_R"].clone(), gt["gt_t"].clone(), gt["gt_s"].clone(),
                                         gt["mean_shape"], gt["sym"], gt["aug_bb"], gt["aug_rt_t"], gt["aug_rt_r"],
                                         gt["model_point"].clone(), gt["nocs_scale"], gt["obj_id"])
    aug.update({f"{tag}.PC": PC.numpy(), f"{tag}.R": R.numpy(), f"{tag}.t": t.numpy(), f"{tag}.s": s.numpy()})
    print(f"  aug {tag}: |dPC| max {float((PC - gt['PC']).abs().max()):.4f}")
save("losses_augment", green=g.numpy(), red=r.numpy(), **aug)
print("full step: HSPose.forward(do_loss=True), closed-form weights, B=4 N=256")
import network.fs_net_repo.gcn3d as rg

torch.set_num_threads(1)      # bit-reproducible regeneration (threaded MKL reductions are not)
B, N, seed = 4, 256, 73
FLAGS.train = 1
FLAGS.aug_bb_pro = FLAGS.aug_rt_pro = FLAGS.aug_bc_pro = FLAGS.aug_pc_pro = -1.0       # augmentation never fires
net = RH.HSPose('PoseNet_only')
sd = net.posenet.state_dict()
oc.fill_state_closed_form(sd)
net.train()
for mod in net.modules():
    if isinstance(mod, torch.nn.Dropout):
        mod.p = 0.0
case = oc.hspose_train_case(B, N, seed)
# record what the GPU test has to replay: the feature-space neighbour sets (selection discontinuity) and the
# Pool_layer draws (they come after the augmentation's draws on the CPU generator here, on a GPU they do not)
feat_idx, perms = [], []
orig_knn, orig_perm = rg.get_neighbor_index, torch.randperm


def rec_knn(vertices, neighbor_num):
    o = orig_knn(vertices, neighbor_num)
    if vertices.shape[-1] != 3:
        feat_idx.append(o.clone())
    return o


def rec_perm(n, *a, **k):
    o = orig_perm(n, *a, **k)
    perms.append(o.clone())
    return o


rg.get_neighbor_index, torch.randperm = rec_knn, rec_perm
torch.manual_seed(1)
out_dict, ld = net(PC=case["PC"], obj_id=case["obj_id"], gt_R=case["gt_R"], gt_t=case["gt_t"], gt_s=case["gt_s"],
                   mean_shape=case["mean_shape"], sym=case["sym"], aug_bb=case["aug_bb"], aug_rt_t=case["aug_rt_t"],
                   aug_rt_r=case["aug_rt_r"], model_point=case["model_point"], nocs_scale=case["nocs_scale"], do_loss=True)
rg.get_neighbor_index, torch.randperm = orig_knn, orig_perm
assert len(feat_idx) == 4 and len(perms) == 2
out = {"meta": np.array([B, N, seed], np.int64)}
for li, fi in enumerate(feat_idx):
    out[f"featknn{li + 1}"] = fi.numpy().astype(np.int16)
out["pool_idx0"] = perms[0][:N // 4].numpy().astype(np.int16)
out["pool_idx1"] = perms[1][:(N // 4) // 4].numpy().astype(np.int16)
for k in ("p_green_R", "p_red_R", "f_green_R", "f_red_R", "Pred_T", "Pred_s"):
    out["out." + k] = out_dict[k].detach().numpy()
n = 0
for grp, d in ld.items():
    for k, v in d.items():
        out[f"{grp}.{k}"] = np.asarray(v.detach().numpy(), np.float32).reshape(-1)
        n += 1
assert n == 19
total = sum(ld['fsnet_loss'].values()) + sum(ld['recon_loss'].values()) + sum(ld['geo_loss'].values()) + sum(ld['prop_loss'].values())
out["total"] = np.array([float(total)], np.float32)
total.backward()
for k_, prm in net.posenet.named_parameters():
    if prm.grad is not None and (k_.endswith("conv_0.directions") or k_.endswith("ts.conv4.weight") or k_.endswith("rot_green.conv1.weight")
                                 or k_.endswith("face_head.9.weight") or k_.endswith("conv_4.weights")):
        out["gradnorm." + k_] = np.array([prm.grad.double().norm().item()])
print(f"  19 terms, total {float(total):.6f}; grad norms kept: {[k for k in out if k.startswith('gradnorm')]}")
save("losses_full_step", **out)
print("done")
