"""Diagnostic (not a test): how far the CPU ORACLE's own HS-stack gradients move under 1-ulp input noise
(same forced feature-KNN sets).  Justifies the full-stack gradient tolerance in tests/test_gpu_stack.py."""
import sys, os
sys.path[:0] = ['/root/repo/oracle', '/root/repo/tests', '/root/repo']
import numpy as np, torch
import ref_cpu as oc
from conftest import golden
name = sys.argv[1]
g = golden(name)
train_flag, B, N, seed, bn_training = (int(v) for v in g["meta"])
torch.set_num_threads(8)
# build closed-form state via the product module (same key set)
from hs_pose_amd.config import FLAGS
FLAGS.train = 0
from hs_pose_amd.FaceRecon import FaceRecon
net = FaceRecon(); sd = net.state_dict(); 
# use posenet-style keys ordering: fill exactly like tests do (on PoseNet9D) for comparable numbers
from hs_pose_amd.PoseNet9D import PoseNet9D
FLAGS.train = train_flag
pn = PoseNet9D(); psd = pn.state_dict(); oc.fill_state_closed_form(psd)
base = {k[len("face_recon."):]: v.detach().clone() for k, v in psd.items() if k.startswith("face_recon.")}
obj = torch.from_numpy((oc.hash_unit(B, seed + 1) * 6).astype(np.int64)).float().view(B, 1)
lists = [torch.from_numpy(g[f"featknn{i}"].astype(np.int64)) for i in (1,2,3,4)]
orig_knn = oc.knn_index
def run(centred):
    calls = [0]
    def forced(x, k):
        if x.shape[-1] == 3: return orig_knn(x, k)
        calls[0] += 1; return lists[calls[0]-1]
    oc.knn_index = forced
    p = {k: v.clone() for k, v in base.items()}
    for k in p:
        if p[k].is_floating_point() and "running" not in k: p[k].requires_grad_(True)
    pidx = [torch.from_numpy(g["pool_idx0"].astype(np.int64)), torch.from_numpy(g["pool_idx1"].astype(np.int64))]
    out = oc.face_recon(p, centred, obj, pidx, train_heads=False, bn_training=True)["feat"]
    dfeat = oc.hash_tensor(tuple(out.shape), seed + 5, 1.0)
    (out * dfeat).sum().backward()
    oc.knn_index = orig_knn
    return out.detach(), {k: v.grad for k, v in p.items() if v.grad is not None}
c0 = torch.from_numpy(g["centred"])
f0, g0 = run(c0)
noise = (oc.hash_tensor(tuple(c0.shape), 999, 1.0)) * 1.2e-7
f1, g1 = run(c0 * (1 + noise))
print("feat change", (f1-f0).abs().max().item())
for k in g0:
    if k.split(".")[0] in ("conv1d_block","recon_head","face_head"): continue
    a, b = g0[k], g1[k]
    print(f"{k:28s} maxerr {(a-b).abs().max().item():.3e} (max {a.abs().max().item():.3e}) normrel {abs(a.norm().item()-b.norm().item())/a.norm().item():.2e}")
