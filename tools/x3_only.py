import sys, os
sys.path.insert(0, "/root/repo")
import torch
from hs_pose_amd import ops
dev = torch.device("cuda:0")
B, S = 16, 7
for name, N, Cin, C in [("conv_1", 1028, 128, 128), ("conv_3", 257, 256, 256)]:
    M = B * N
    X = torch.randn(M, Cin, device=dev); W = torch.randn(Cin, (S + 1) * C, device=dev) * 0.05; b = torch.randn((S + 1) * C, device=dev)
    wste = torch.randn(C, Cin, device=dev) * 0.05
    g2 = torch.randn(M, C, device=dev); gfm = torch.randn(M, (S + 1) * C, device=dev)
    gX = torch.empty(M, Cin, device=dev); fm = torch.empty(M, (S + 1) * C, device=dev)
    for _ in range(10):
        ops._fm_rows(X, W, b, out=fm)
        ops._grad_in_rows(g2, wste, gfm, W, gX)
torch.cuda.synchronize()
