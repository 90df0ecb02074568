"""diagnostic: per-layer deviation of the bf16 path from the fp32 path (same weights, same neighbour lists)"""
import os, sys
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
net = FaceRecon().to(dev).train()
net.set_feature_dtype(torch.bfloat16)
net._bf16.refresh()
B, N, k, S = 2, 1028, 20, 7
xyz = oc.hash_tensor((B, N, 3), 61, 0.05).to(dev)
xyz = xyz - xyz.mean(dim=1, keepdim=True)
idx_x = ops.knn(xyz, k)

def rel(a, b):
    a, b = a.float(), b.float()
    return f"max {((a - b).abs().max() / b.abs().max()).item():.2e} rms {((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item():.2e}"

c0 = net.conv_0
of = ops.surface_layer(xyz, idx_x, k, S, c0.directions, c0.STE_layer.weight, c0.conv2.weight)
ob = ops_bf16.surface_layer(xyz, idx_x, k, S, c0.directions, c0.STE_layer.weight, c0.conv2.weight)
print("conv_0 out:", rel(ob, of), " |out| rms", of.pow(2).mean().sqrt().item(), "max", of.abs().max().item())
X = torch.relu(of)
for name, Xin in (("conv_1 (same bf16-rounded input)", X.bfloat16()),):
    l = net.conv_1
    idx_f = ops.knn(Xin.float(), k)
    of1 = ops.hs_layer(xyz, Xin.float(), idx_f, idx_x, k, S, l.weights, l.bias, l.directions, l.STE_layer.weight, l.conv2.weight)
    ob1 = ops_bf16.hs_layer(xyz, Xin, idx_f, idx_x, k, S, l.weights, l.bias, l.directions, l.STE_layer.weight, l.conv2.weight)
    print(name, "out:", rel(ob1, of1), " |out| rms", of1.pow(2).mean().sqrt().item(), "max", of1.abs().max().item())
    # pieces
    X2 = Xin.view(B * N, -1)
    W_b, WT_b = ops_bf16.copies_of(l.weights)
    fm_b = ops.gemm_rows(X2, WT_b, bias=l.bias)
    fm_f = torch.addmm(l.bias, X2.float(), l.weights)
    print("   fm:", rel(fm_b, fm_f))
    fm_f2 = torch.addmm(l.bias, X2.float(), W_b.float())
    print("   fm vs fp32 product of the SAME bf16 weights:", rel(fm_b, fm_f2))
    bn = net.bn1
    yb = ops.bn_relu(ob1, bn); yf = ops.bn_relu(of1, bn)
    print("   after bn1+relu:", rel(yb, yf), " input mean/std per channel ratio (median)", (of1.mean(dim=(0, 1)).abs() / of1.std(dim=(0, 1))).median().item())
