"""Diagnostic (not a test): HS-stack parameter-gradient error table vs the golden fixture."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import ref_cpu as ref
from conftest import golden
from hs_pose_amd.config import FLAGS
from hs_pose_amd import ops
from hs_pose_amd.PoseNet9D import PoseNet9D
dev = torch.device("cuda:0")
name = sys.argv[1]
g = golden(name)
train_flag, B, N, seed, bn_training = (int(v) for v in g["meta"])
FLAGS.train = train_flag
net = PoseNet9D(); ref.fill_state_closed_form(net.state_dict()); net = net.to(dev).train(True)
obj = torch.from_numpy((ref.hash_unit(B, seed + 1) * 6).astype(np.int64)).float().view(B, 1).to(dev)
real = ops.knn
lists = [torch.from_numpy(g[f"featknn{i}"].astype(np.int32)).to(dev) for i in (1, 2, 3, 4)]
calls = [0]
def forced(x, k, drop_first=True):
    if x.shape[-1] == 3: return real(x, k, drop_first)
    calls[0] += 1
    return lists[(calls[0] - 1) % 4]
ops.knn = forced
torch.manual_seed(1)
_, _, feat = net.face_recon(torch.from_numpy(g["centred"]).to(dev), obj)
dfeat = ref.hash_tensor(tuple(feat.shape), seed + 5, 1.0).to(dev)
(feat * dfeat).sum().backward()
for pn, p in net.face_recon.named_parameters():
    key = "gradsample." + pn
    if key not in g.files: continue
    want = g[key]; norm, _ = g["gradnorm." + pn]
    got = p.grad.reshape(-1)[::499].cpu().double().numpy()
    gn = p.grad.double().norm().item()
    print(f"{pn:28s} sample maxerr {np.abs(got-want).max():.3e} (max|want| {np.abs(want).max():.3e})  norm {gn:.6e} vs {norm:.6e} rel {abs(gn-norm)/norm:.2e}")
