"""diagnostic: per-level deviation of the bf16 HS stack from the fp32 path (same weights, pool draws, replayed feature-KNN)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import ref_cpu as oc
from hs_pose_amd import ops
from hs_pose_amd.config import FLAGS
from hs_pose_amd.FaceRecon import FaceRecon
dev = torch.device("cuda:0")
FLAGS.reset(); FLAGS.train = 0
nets = []
for dt in (torch.float32, torch.bfloat16):
    torch.manual_seed(0)
    n = FaceRecon().to(dev).train(); n.set_feature_dtype(dt); nets.append(n)
B, N = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 1028
pts = oc.hash_tensor((B, N, 3), 61, 0.05).to(dev); pts = pts - pts.mean(dim=1, keepdim=True)
obj = torch.tensor([[1.0], [4.0]]).to(dev)
real = ops.knn; lists = []; mode = ["rec"]; pos = [0]
def knn(x, k, drop_first=True, **kw):
    own = real(x, k, drop_first)
    if x.shape[-1] == 3: return own
    if mode[0] == "rec": lists.append(own); return own
    w = lists[pos[0]]; pos[0] += 1; return w
ops.knn = knn
torch.manual_seed(5); _, _, ff = nets[0](pts, obj)
mode[0] = "rep"
torch.manual_seed(5); _, _, fb = nets[1](pts, obj)
segs = [("fm_0 relu(conv_0)", 0, 128), ("fm_1 bn1", 128, 256), ("fm_2 bn2 (up)", 256, 512), ("fm_3 bn3 (up)", 512, 768), ("fm_4 conv_4 (up)", 768, 1280)]
for name, a, b in segs:
    x, y = fb[:, :, a:b].float(), ff[:, :, a:b]
    print(f"{name:22s} max err {((x - y).abs().max() / y.abs().max()).item():.2e} of max   rms err {((x - y).pow(2).mean().sqrt() / y.pow(2).mean().sqrt()).item():.2e} of rms"
          f"   fraction of entries off by > 2% of max: {((x - y).abs() > 0.02 * y.abs().max()).float().mean().item():.2e}")
