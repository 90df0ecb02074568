"""GPU box: where does the eval-mode forward stop being bit-identical to the CPU oracle (= the reference's arithmetic)?
Prints the fraction of bit-equal elements stage by stage."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import torch.nn.functional as F
import ref_cpu as oc
from hs_pose_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
print("host threads", torch.get_num_threads(), "-> 1 (the fixtures are written with one thread: threaded MKL splits K)")
Xp = torch.relu(torch.randn(2056, 256)); Wp = torch.randn(256, 2048) * 0.05
r_many = Xp @ Wp
torch.set_num_threads(1)
print("X W (K = 256) with the host's default threads == with one thread:", float((r_many == Xp @ Wp).float().mean()))


def eqf(a, b, nm):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    eq = (a == b).float().mean().item()
    print(f"{nm:34s} equal {eq:.6f}   max abs diff {(a - b).abs().max().item():.3e}")
    return eq


B, N, C, k, S = 2, 1028, 128, 20, 7
xyz = torch.randn(B, N, 3) * 0.05
D = (torch.rand(3, S * C) * 2 - 1) / (S * C) ** 0.5
idx = oc.knn_index(xyz, k)
idx_g = ops.knn(xyz.to(dev), k)
assert torch.equal(idx_g.cpu().long(), idx)
# --- surface graph conv: unit directions, theta chain, max, mean over supports
want = oc.surface_graph_conv(oc.neighbor_dirs(xyz, idx), D, S, C)
got = ops.rf_surface(xyz.to(dev), idx_g, D.to(dev), S)
eqf(got, want, "surface graph conv")
# --- products
for (M, K, Nn) in ((B * N, 128, 1024), (B * N, 256, 2048), (B * N, 128, 128), (B * N, 256, 256), (16, 256, 256)):
    X = torch.relu(torch.randn(M, K)); W = torch.randn(K, Nn) * 0.05; b = torch.randn(Nn) * 0.1
    eqf(ops.gemm_wave(X.to(dev), W.to(dev), True, bias=b.to(dev)), X @ W + b, f"X W + b  M{M} K{K} N{Nn}")
    Wt = torch.randn(Nn, K) * 0.05
    eqf(ops.gemm_wave(X.to(dev), Wt.to(dev), False), oc._conv1x1(X.unsqueeze(0), Wt.unsqueeze(-1)).squeeze(0), f"conv1d   M{M} K{K} N{Nn}")
# --- ORL mean
for (n, c) in ((1028, 128), (257, 256), (64, 512), (300, 128)):
    f3 = torch.randn(B, n, c)
    x3 = torch.randn(B, n, 3) * 0.05
    kk = min(20, n // 8)
    want = oc.orl_global(f3, x3, kk)[:, 0]
    got, _ = ops._orl_fwd_exact(f3.to(dev), ops.knn(x3.to(dev), kk), kk)
    eqf(got, want, f"ORL mean N{n} C{c}")
# --- eval BatchNorm
bn = torch.nn.BatchNorm1d(128).eval()
bn.running_mean.normal_(); bn.running_var.uniform_(0.2, 3); bn.weight.data.normal_(); bn.bias.data.normal_()
x = torch.randn(B, N, 128)
with torch.no_grad():
    want = torch.relu(bn(x.transpose(1, 2)).transpose(1, 2))
    got = ops.bn_relu(x.to(dev), bn.to(dev), relu=True)
eqf(got, want, "eval BatchNorm + relu")
bn = bn.cpu()
# --- whole layers
p = {"directions": D, "STE_layer.weight": torch.randn(C, 3, 1) * 0.3, "conv2.weight": torch.randn(C, 2 * C, 1) * 0.05}
with torch.no_grad():
    want = oc.surface_layer({"c." + k_: v for k_, v in p.items()}, "c.", xyz, k, S)
    with ops.exact_scope(True):
        got = ops.surface_layer(xyz.to(dev), idx_g, k, S, D.to(dev), p["STE_layer.weight"].to(dev), p["conv2.weight"].to(dev))
    fm0 = torch.relu(want)
    eqf(got, want, "surface layer (conv_0)")
    for (Cin, Co, n) in ((128, 128, 1028), (128, 256, 257), (256, 256, 257)):
        kk = min(20, n // 8) if n < 1028 else 20
        xz = torch.randn(B, n, 3) * 0.05
        Xf = torch.relu(torch.randn(B, n, Cin))
        q = {"weights": (torch.rand(Cin, (S + 1) * Co) * 2 - 1) * 0.03, "bias": (torch.rand((S + 1) * Co) * 2 - 1) * 0.03,
             "directions": (torch.rand(3, S * Co) * 2 - 1) * 0.03, "STE_layer.weight": torch.randn(Co, Cin, 1) * 0.05,
             "conv2.weight": torch.randn(Co, 2 * Co, 1) * 0.05}
        want, idf = oc.hs_layer({"c." + k_: v for k_, v in q.items()}, "c.", xz, Xf, kk, S, return_idx=True)
        idf_g = ops.knn(Xf.to(dev), kk)
        print("   feature KNN lists equal:", torch.equal(idf_g.cpu().long(), idf))
        with ops.exact_scope(True):
            got = ops.hs_layer(xz.to(dev), Xf.to(dev), idf_g, ops.knn(xz.to(dev), kk), kk, S, q["weights"].to(dev), q["bias"].to(dev),
                               q["directions"].to(dev), q["STE_layer.weight"].to(dev), q["conv2.weight"].to(dev))
        eqf(got, want, f"HS layer {Cin}->{Co} N{n}")
        # its graph conv alone
        fm = (Xf @ q["weights"] + q["bias"])
        want_gc = oc.hs_graph_conv(oc.neighbor_dirs(xz, idf), idf, Xf, q["weights"], q["bias"], q["directions"], S)
        got_gc = ops.rf_conv(xz.to(dev), idf_g, q["directions"].to(dev), fm.to(dev), S)
        eqf(got_gc, want_gc, "   its graph conv (fm from CPU)")
