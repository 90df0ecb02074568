"""diagnostic: which part of the bf16 step breaks hipGraph capture"""
import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import ref_cpu as oc
from hs_pose_amd import ops, ops_bf16, gcn3d
from hs_pose_amd.config import FLAGS
from hs_pose_amd.FaceRecon import FaceRecon
dev = torch.device("cuda:0")
FLAGS.reset(); FLAGS.train = 0
torch.manual_seed(0)
net = FaceRecon().to(dev).train(); net.set_feature_dtype(torch.bfloat16)
B, N, k, S = 2, 256, 20, 7
xyz = oc.hash_tensor((B, N, 3), 61, 0.05).to(dev)
obj = torch.tensor([[1.0], [4.0]]).to(dev)
dfeat = oc.hash_tensor((B, N, 1286), 62, 1.0).to(dev).bfloat16()
which = sys.argv[1]

def body():
    if which == "refresh":
        net._bf16.refresh()
    elif which == "conv0":
        net._bf16.refresh()
        with gcn3d.knn_scope():
            return net.conv_0(xyz, k)
    elif which == "conv01":
        net._bf16.refresh()
        with gcn3d.knn_scope():
            f0 = torch.relu(net.conv_0(xyz, k))
            return net.conv_1(xyz, f0, k)
    elif which == "fwd":
        with torch.no_grad():
            return net(xyz, obj)[2]
    elif which == "fwdbwd":
        for p in net.parameters(): p.grad = None
        f = net(xyz, obj)[2]
        f.backward(dfeat)
    elif which == "conv0bwd":
        net._bf16.refresh()
        for p in net.parameters(): p.grad = None
        with gcn3d.knn_scope():
            o = net.conv_0(xyz, k)
        o.backward(torch.ones_like(o))
    elif which == "conv01bwd":
        net._bf16.refresh()
        for p in net.parameters(): p.grad = None
        with gcn3d.knn_scope():
            f0 = torch.relu(net.conv_0(xyz, k))
            o = net.conv_1(xyz, f0, k)
        o.backward(torch.ones_like(o))

side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): body()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    body()
g.replay(); torch.cuda.synchronize()
print(which, "captured + replayed OK")
