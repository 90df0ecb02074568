"""GPU box (debug): layer outputs of the reference-initialised eval stack against the full dump of
oracle/gen_golden_exact.py --debug-dump build_tmp/exact_debug.npz"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import ref_cpu as ref
from hs_pose_amd.config import FLAGS
from hs_pose_amd.PoseNet9D import PoseNet9D
d = np.load(os.path.join(ROOT, "build_tmp", "exact_debug.npz"))
dev = torch.device("cuda:0")
FLAGS.reset(); FLAGS.train = 0
torch.manual_seed(0)
net = PoseNet9D().to(dev).eval()
fr = net.face_recon
B, N, seed = 2, 1028, 81
pts = ref.hash_tensor((B, N, 3), seed, 0.05); pts[:, :, 2] += 0.8
obj = torch.from_numpy((ref.hash_unit(B, seed + 1) * 6).astype(np.int64)).float().view(B, 1)
pts = pts - pts.mean(dim=1, keepdim=True)
grabbed = {}
for nm in ("conv_0", "conv_1", "conv_2", "conv_3", "conv_4"):
    getattr(fr, nm).register_forward_hook(lambda mod, i, o, nm=nm: grabbed.__setitem__(nm, (o[0] if isinstance(o, tuple) else o).detach().cpu().numpy()))
torch.manual_seed(1)
with torch.no_grad():
    _, _, feat = fr(pts.to(dev), obj.to(dev))
for nm in ("conv_0", "conv_1", "conv_2", "conv_3", "conv_4"):
    a, b = grabbed[nm], d[nm]
    print(nm, "equal", float((a == b).mean()), "max abs diff", float(np.abs(a - b).max()), "rows fully equal", float((a == b).all(-1).mean()))
f = feat.cpu().numpy()[..., :1286]
print("feat equal", float((f == d["feat"]).mean()), float(np.abs(f - d["feat"]).max()))
# --- conv_1 in isolation on the reference's own fm_0
from hs_pose_amd import ops, gcn3d
g = np.load(os.path.join(ROOT, "tests", "golden", "stack_refinit_eval_1028.npz"))
fm0 = torch.relu(torch.from_numpy(d["conv_0"])).to(dev)
idx = ops.knn(fm0, 20)
want_idx = torch.from_numpy(g["featknn1"].astype(np.int64))
print("feature KNN on the reference's fm_0: rows with the reference's ordered list", float((idx.cpu().long() == want_idx).all(-1).float().mean()))
oidx = ref.knn_index(fm0.cpu(), 20)
print("   CPU oracle (this host) vs fixture:", float((oidx == want_idx).all(-1).float().mean()), " GPU vs CPU oracle (this host):", float((idx.cpu().long() == oidx).all(-1).float().mean()))
xyz = pts.to(dev)
with torch.no_grad(), ops.exact_scope(True), gcn3d.knn_scope():
    out = fr.conv_1(xyz, fm0, 20)
a, b = out.cpu().numpy(), d["conv_1"]
print("conv_1 on the reference's fm_0: equal", float((a == b).mean()), "max abs diff", float(np.abs(a - b).max()))
with torch.no_grad(), ops.exact_scope(True), gcn3d.knn_scope():
    out = ops.hs_layer(xyz, fm0, want_idx.int().to(dev), ops.knn(xyz, 20), 20, 7, fr.conv_1.weights, fr.conv_1.bias, fr.conv_1.directions,
                       fr.conv_1.STE_layer.weight, fr.conv_1.conv2.weight)
a = out.cpu().numpy()
print("   ... with the reference's neighbour lists: equal", float((a == b).mean()), "max abs diff", float(np.abs(a - b).max()))
