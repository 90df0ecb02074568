"""GPU box (debug): rows where the feature-space KNN differs from the CPU oracle on the reference's fm_0"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import ref_cpu as ref
from hs_pose_amd import ops
torch.set_num_threads(1)
d = np.load(os.path.join(ROOT, "build_tmp", "exact_debug.npz"))
dev = torch.device("cuda:0")
fm0 = torch.relu(torch.from_numpy(d["conv_0"]))
idx = ops.knn(fm0.to(dev), 20).cpu().long()
oidx = ref.knn_index(fm0, 20)
inner = torch.bmm(fm0, fm0.transpose(1, 2)); quad = torch.sum(fm0 ** 2, dim=2)
dist = inner * (-2) + quad.unsqueeze(1) + quad.unsqueeze(2)
bad = (idx != oidx).any(-1)
print("rows differing:", int(bad.sum()), "of", bad.numel(), " zero fraction of fm0:", float((fm0 == 0).float().mean()))
print("duplicate rows in fm0:", int(fm0.shape[1] * fm0.shape[0] - sum(len(torch.unique(fm0[b], dim=0)) for b in range(fm0.shape[0]))))
n = 0
for b in range(fm0.shape[0]):
    for i in torch.nonzero(bad[b]).flatten().tolist()[:6]:
        g_, o_ = idx[b, i], oidx[b, i]
        pos = torch.nonzero(g_ != o_).flatten().tolist()
        print(f"b{b} row {i}: first differing slot {pos[0]} (of {len(pos)}): gpu {g_[pos[0]].item()} d={dist[b, i, g_[pos[0]]].item():.9g}  cpu {o_[pos[0]].item()} d={dist[b, i, o_[pos[0]]].item():.9g}"
              f"   same SET: {sorted(g_.tolist()) == sorted(o_.tolist())}; self dist {dist[b, i, i].item():.3g}; rank0 cpu {torch.topk(dist[b, i], 21, largest=False)[1][0].item()}")
# set-level agreement and whether differences are pure ties
same_set = (torch.sort(idx, -1)[0] == torch.sort(oidx, -1)[0]).all(-1)
print("rows with the same neighbour SET:", float(same_set.float().mean()))
dg = torch.gather(dist, 2, idx); do = torch.gather(dist, 2, oidx)
print("rows whose sorted distance lists are identical (differences are exact ties):", float((dg == do).all(-1).float().mean()))
