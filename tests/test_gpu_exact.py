"""The eval-mode forward in the reference's own rounding order: every statement, the HS layers and the
reference-initialised stack against fixtures written by the imported reference (oracle/gen_golden_exact.py) -- BIT FOR BIT, not
to a tolerance.  With the feature rows carrying the reference's bits the feature-space neighbour search selects the reference's
lists, so free-running parity no longer hinges on near ties."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


def _same(got, want, what):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    bad = got != want
    assert not bad.any(), (f"{what}: {int(bad.sum())} of {bad.size} elements differ from the reference's bits "
                           f"(max abs diff {np.abs(got.astype(np.float64) - want).max():.3e})")


def test_exact_statements(dev, ref):
    """products (matmul + bias, Conv1d(k=1), conv2 over cat[F, f_global] with 2C = 256 and 512), the mean over the points, the
    neighbour-direction normalisation (through the surface graph convolution below) and the eval-mode BatchNorm"""
    from hs_pose_amd import ops
    g = golden("exact_statements")
    for K in (128, 256):
        M, N = 96, 64
        X = torch.relu(ref.hash_tensor((M, K), 7001 + K, 1.0)).to(dev)
        W, b = ref.hash_tensor((K, N), 7002 + K, 0.05).to(dev), ref.hash_tensor((N,), 7003, 0.1).to(dev)
        _same(ops.gemm_wave(X, W, True, bias=b), g[f"mm_k{K}"], f"X W + b, K = {K}")
        Wt = ref.hash_tensor((N, K), 7004 + K, 0.05).to(dev)
        _same(ops.gemm_wave(X, Wt, False), g[f"conv_k{K}"], f"Conv1d(k=1), K = {K}")
    for C in (128, 256):
        Fm = ref.hash_tensor((2, 48, C), 7100 + C, 1.0).to(dev)
        fg = ref.hash_tensor((2, 1, C), 7101 + C, 0.5).to(dev).reshape(2, C)
        W2 = ref.hash_tensor((C, 2 * C, 1), 7102 + C, 0.05).to(dev).squeeze(-1)
        out3 = torch.empty(2, 48, C, device=dev)
        zero = torch.zeros(96, C, device=dev)
        ops._layer_out_exact(Fm.view(96, C), W2, fg, 48, out3, ste=zero)          # ((conv2) + F) + 0
        want = (g[f"conv2_c{C}"] + Fm.view(96, C).cpu().numpy()).astype(np.float32)
        _same(out3.view(96, C), want, f"conv2 over cat[F, f_global], 2C = {2 * C}")
    for N in (1028, 257, 64, 300):
        x = ref.hash_tensor((2, N, 32), 7300 + N, 1.0).to(dev)
        idx = torch.arange(N, dtype=torch.int32, device=dev).view(1, N, 1).repeat(2, 1, 1).contiguous()   # "neighbourhood" = the row
        fg, _ = ops._orl_fwd_exact(x, idx, 1)
        _same(fg, g[f"mean_n{N}"], f"mean over {N} points")
    from hs_pose_amd._lib import lib
    from hs_pose_amd.ops import _p, _run, _stream
    for (N, C) in ((257, 256), (100, 64), (1028, 32)):
        x = torch.relu(ref.hash_tensor((2, C, N), 7350 + N, 1.0)).transpose(1, 2).contiguous().to(dev)      # our layout: (B,N,C)
        q = torch.empty(2, N, device=dev)
        _run("hsp_quad_outer_f32", (_p(x), 2, N, C, _p(q), _stream()))
        _same(q, g[f"quad_outer_n{N}"], f"|x|^2 of a transposed view, N = {N}")
    for N in (1028, 100):
        pts = ref.hash_tensor((2, N, 3), 7360 + N, 0.05); pts[:, :, 2] += 0.8
        local, mean = ops.center_cloud(pts.to(dev))
        _same(mean, g[f"centre_mean_n{N}"], f"cloud mean, N = {N}")
        _same(local, g[f"centre_local_n{N}"], f"centred cloud, N = {N}")
    C = 256
    bn = torch.nn.BatchNorm1d(C).eval()
    with torch.no_grad():
        bn.running_mean.copy_(ref.hash_tensor((C,), 7500, 1.0)); bn.running_var.copy_(ref.hash_tensor((C,), 7501, 1.2).abs() + 0.2)
        bn.weight.copy_(ref.hash_tensor((C,), 7502, 1.0)); bn.bias.copy_(ref.hash_tensor((C,), 7503, 1.0))
        bn = bn.to(dev)
        x = ref.hash_tensor((2, 60, C), 7504, 1.5).to(dev)
        inv = ops._eval_invstd(bn)
        if np.array_equal(inv.cpu().numpy(), g["bn_invstd"]):              # the host's ATen evaluates 1 / sqrt(v + eps) as the fixture's did
            _same(ops.bn_relu(x, bn, relu=False), g["bn_eval"], "eval BatchNorm")
        else:                                                              # another sqrt on this host: the formula with the fixture's invstd
            bn._hsp_invstd = (bn._hsp_invstd[0], torch.from_numpy(g["bn_invstd"]).to(dev))
            _same(ops.bn_relu(x, bn, relu=False), g["bn_eval"], "eval BatchNorm (fixture invstd)")


def test_exact_layers(dev, ref):
    """HSlayer_surface / HS_layer (2C = 256: one conv2 chain; 2C = 512: two) in eval mode: the reference's output bits"""
    from hs_pose_amd import gcn3d, ops
    g = golden("exact_layers")
    S, k = 7, 20
    xyz = ref.hash_tensor((2, 128, 3), 7600, 0.05).to(dev)
    m = gcn3d.HSlayer_surface(kernel_num=128, support_num=S).eval()
    for i, (kk, v) in enumerate(m.state_dict().items()):
        ref.hash_fill_(v, 7610 + i, 0.3 if "STE" in kk else 0.05)
    m = m.to(dev)
    with torch.no_grad(), ops.exact_scope(True), gcn3d.knn_scope():
        _same(m(xyz, k), g["surface"], "HSlayer_surface")
    for tag, (Cin, Co, n, kk_) in {"hs128": (128, 128, 128, 20), "hs256": (128, 256, 96, 12), "hs256b": (256, 256, 96, 12)}.items():
        m = gcn3d.HS_layer(Cin, Co, support_num=S).eval()
        for i, (kk, v) in enumerate(m.state_dict().items()):
            ref.hash_fill_(v, 7700 + 10 * Cin // 128 + 100 * Co // 128 + i, 0.05)
        m = m.to(dev)
        X = torch.relu(ref.hash_tensor((2, n, Cin), 7800 + Cin + Co, 1.0)).to(dev)
        with torch.no_grad(), ops.exact_scope(True), gcn3d.knn_scope():
            _same(m(xyz[:, :n].contiguous(), X, kk_), g[tag], f"HS_layer {Cin} -> {Co}")


def test_exact_stack_refinit(dev, ref, flags):
    """the reference-initialised HS stack, eval mode, N = 1028, free-running: conv_0 ... conv_3 outputs equal the reference's bits
    (so every feature-space neighbour list is the reference's), conv_4 and feat to 2e-6 (conv_4's conv2 spans 1024 channels:
    four block chains on the CPU, and its rows rank nothing)"""
    from hs_pose_amd.PoseNet9D import PoseNet9D
    g = golden("exact_stack_1028")
    B, N, seed, _ = (int(v) for v in g["meta"])
    flags.train = 0
    torch.manual_seed(0)
    net = PoseNet9D().to(dev).eval()
    fr = net.face_recon
    pts = ref.hash_tensor((B, N, 3), seed, 0.05)
    pts[:, :, 2] += 0.8
    obj = torch.from_numpy((ref.hash_unit(B, seed + 1) * 6).astype(np.int64)).float().view(B, 1)
    grabbed, hooks = {}, []
    for nm in ("conv_0", "conv_1", "conv_2", "conv_3", "conv_4"):
        hooks.append(getattr(fr, nm).register_forward_hook(
            lambda mod, i, o, nm=nm: grabbed.__setitem__(nm, (o[0] if isinstance(o, tuple) else o).detach().clone())))
    from hs_pose_amd import ops
    torch.manual_seed(1)
    with torch.no_grad():
        local, _ = ops.center_cloud(pts.to(dev))                 # PoseNet9D.py:25, the mean in the reference's order
        _, _, feat = fr(local, obj.to(dev))
    for h_ in hooks:
        h_.remove()
    report = []
    for nm in ("conv_0", "conv_1", "conv_2", "conv_3"):
        got = grabbed[nm].reshape(-1)[::53].cpu().numpy()
        report.append((nm, float((got == g[nm]).mean()), float(np.abs(got - g[nm]).max())))
    print("EXACT STACK: fraction of sampled elements with the reference's bits / max abs diff:", report)
    for nm, eq, _ in report:
        assert eq == 1.0, report
    got4 = grabbed["conv_4"].reshape(-1)[::53].cpu().numpy()
    assert np.abs(got4 - g["conv_4"]).max() <= 2e-6 * max(1.0, np.abs(g["conv_4"]).max())
    gotf = feat[..., :1286].reshape(-1)[::211].cpu().numpy() if feat.shape[-1] != 1286 else feat.reshape(-1)[::211].cpu().numpy()
    assert np.abs(gotf - g["feat"]).max() <= 2e-6 * max(1.0, np.abs(g["feat"]).max())
    # the whole PoseNet9D, free-running, against the reference's pose / size outputs and neighbour lists (the refinit fixture)
    g2 = golden("stack_refinit_eval_1028")
    lists = []
    real_knn = ops.knn

    def rec(x, k, *a, **kw):
        o = real_knn(x, k, *a, **kw)
        if x.shape[-1] != 3:
            lists.append(o.cpu().numpy())
        return o
    ops.knn = rec
    try:
        torch.manual_seed(1)
        with torch.no_grad():
            outs = net(pts.to(dev), obj.to(dev))
    finally:
        ops.knn = real_knn
    agree = [float((l_ == g2[f"featknn{i + 1}"]).all(-1).mean()) for i, l_ in enumerate(lists[:4])]
    names = ["p_green_R", "p_red_R", "f_green_R", "f_red_R", "Pred_T", "Pred_s"]
    errs = {n_: float(np.abs(o.cpu().numpy() - g2["out." + n_]).max()) for n_, o in zip(names, outs[4:])}
    print(f"EXACT FREE-RUNNING PoseNet9D (eval, N = 1028): rows with the reference's ordered neighbour list per HS layer {agree}; "
          f"max abs error of the pose / size outputs {errs}")
    assert agree == [1.0, 1.0, 1.0, 1.0], agree
    assert max(errs.values()) <= 1e-5, errs


def test_exact_stack_tiled(dev, ref, flags, monkeypatch):
    """TILED clouds (a 400-point and a 1000-point crop padded to 1028 by repetition, datasets/load_data.py:314-316) through the
    reference-initialised network in eval mode, free-running, against the imported reference (oracle/gen_golden_tiled.py): every
    ordered neighbour list -- the coordinate searches at k = 20 AND Pool_layer's own k = 4 (gcn3d.py:236), which on such a cloud
    is not the prefix of the k = 20 list on half of the rows, and the four feature-space searches, where the duplicated points'
    identical feature rows tie everywhere --, both Pool_layer outputs and conv_0 ... conv_3 bit for bit, the six outputs to 1e-5."""
    from conftest import tiled_batch
    from hs_pose_amd import ops
    from hs_pose_amd.PoseNet9D import PoseNet9D
    g = golden("exact_stack_tiled_1028")
    B, N, seed = (int(v) for v in g["meta"][:3])
    bases = [int(v) for v in g["meta"][4:]]
    flags.train = 0
    torch.manual_seed(0)
    net = PoseNet9D().to(dev).eval()
    fr = net.face_recon
    pts = tiled_batch(ref, bases, seed, N)
    obj = torch.from_numpy((ref.hash_unit(B, seed + 1) * 6).astype(np.int64)).float().view(B, 1)
    grabbed, hooks = {}, []
    for nm in ("conv_0", "conv_1", "conv_2", "conv_3", "conv_4", "pool_1", "pool_2"):
        hooks.append(getattr(fr, nm).register_forward_hook(          # (cloned: FaceRecon applies relu to conv_0's output in place)
            lambda mod, i, o, nm=nm: grabbed.__setitem__(nm, tuple(t.detach().clone() for t in o) if isinstance(o, tuple) else o.detach().clone())))
    xyz_lists, feat_lists = {}, []
    real_xyz, real_knn = ops.knn_xyz, ops.knn

    def rec_xyz(x, k, k2=0, drop_first=True):
        a, b = real_xyz(x, k, k2, drop_first)
        xyz_lists[(x.shape[1], k)] = a
        if b is not None:
            xyz_lists[(x.shape[1], k2)] = b
        return a, b

    def rec_knn(x, k, *a, **kw):
        o = real_knn(x, k, *a, **kw)
        if x.shape[-1] != 3:
            feat_lists.append(o)
        return o
    monkeypatch.setattr(ops, "knn_xyz", rec_xyz)
    real_geo = ops.geometry_levels

    def rec_geo(xyz, sel1, sel2, k1, kpool, k2):             # (the two coarse levels' lists come out of one fused launch)
        geo = real_geo(xyz, sel1, sel2, k1, kpool, k2)
        if geo is not None:
            xyz_lists[(sel1.numel(), k1)], xyz_lists[(sel1.numel(), kpool)], xyz_lists[(sel2.numel(), k2)] = geo["idx1"], geo["idx1_pool"], geo["idx2"]
        return geo
    monkeypatch.setattr(ops, "geometry_levels", rec_geo)
    real_all = ops.geometry_all

    def rec_all(xyz, k0, kpool0, sel1, sel2, k1, kpool, k2):   # (... and, where the shapes allow, the input cloud's lists with them)
        geo = real_all(xyz, k0, kpool0, sel1, sel2, k1, kpool, k2)
        if geo is not None:
            xyz_lists[(xyz.shape[1], k0)], xyz_lists[(xyz.shape[1], kpool0)] = geo["idx0"], geo["idx0_pool"]
            xyz_lists[(sel1.numel(), k1)], xyz_lists[(sel1.numel(), kpool)], xyz_lists[(sel2.numel(), k2)] = geo["idx1"], geo["idx1_pool"], geo["idx2"]
        return geo
    monkeypatch.setattr(ops, "geometry_all", rec_all)
    monkeypatch.setattr(ops, "knn", rec_knn)
    torch.manual_seed(1)
    with torch.no_grad():
        outs = net(pts.to(dev), obj.to(dev))
    for h_ in hooks:
        h_.remove()
    # every coordinate search the reference ran, list for list
    for (n_, k_) in ((1028, 20), (1028, 4), (257, 20), (257, 4), (64, 8)):
        got, want = xyz_lists[(n_, k_)].cpu().numpy(), g[f"xyz_n{n_}_k{k_}"]
        rows = float((got == want).all(-1).mean())
        assert rows == 1.0, f"xyz search N = {n_}, k = {k_}: {rows:.4f} of the rows carry the reference's ordered list"
    agree = [float((l_.cpu().numpy() == g[f"featknn{i + 1}"]).all(-1).mean()) for i, l_ in enumerate(feat_lists[:4])]
    print(f"EXACT TILED STACK: rows with the reference's ordered feature-space list per HS layer {agree}; "
          f"rows whose k = 4 list differs from the k = 20 prefix {g['k4_vs_k20_prefix'].tolist()}")
    assert agree == [1.0, 1.0, 1.0, 1.0], agree
    for nm in ("pool_1", "pool_2"):
        _same(grabbed[nm][0], g[nm + ".vertices"], nm + " vertices")
        _same(grabbed[nm][1].reshape(-1)[::29], g[nm + ".feature"], nm + " feature")
    for nm in ("conv_0", "conv_1", "conv_2", "conv_3"):
        o = grabbed[nm]
        _same((o[0] if isinstance(o, tuple) else o).reshape(-1)[::53], g[nm], nm)
    got4 = grabbed["conv_4"].reshape(-1)[::53].cpu().numpy()
    assert np.abs(got4 - g["conv_4"]).max() <= 2e-6 * max(1.0, np.abs(g["conv_4"]).max())
    names = ["p_green_R", "p_red_R", "f_green_R", "f_red_R", "Pred_T", "Pred_s"]
    errs = {n_: float(np.abs(o.cpu().numpy() - g["out." + n_]).max()) for n_, o in zip(names, outs[4:])}
    print(f"EXACT TILED STACK: max abs error of the pose / size outputs {errs}")
    assert max(errs.values()) <= 1e-5, errs


@pytest.mark.parametrize("B,N,C,k", [(2, 1028, 128, 20), (3, 257, 256, 20), (2, 64, 256, 8), (2, 1028, 3, 20), (2, 1028, 3, 4),
                                     (1, 300, 32, 12), (2, 4096, 64, 20)])
def test_knn_exact_tie_order(dev, ref, B, N, C, k):
    """rows on a small integer lattice: every distance is an exact integer in fp32 whatever the summation order, and nearly every
    row holds ties among its nearest and at the boundary -- the exact-scope neighbour search returns torch.topk's CPU order
    (ref.knn_index runs the torch ops of gcn3d.py:15-24 on the host), the default search its own lowest-index order"""
    from hs_pose_amd import ops
    g = torch.Generator().manual_seed(N + C + k)
    x = torch.randint(0, 3 if C > 3 else 6, (B, N, C), generator=g).float()
    want = ref.knn_index(x, k)
    with ops.exact_scope(True):
        got = ops.knn(x.to(dev), k)
    rows = (got.cpu().long() == want).all(-1).float().mean().item()
    plain = (ops.knn(x.to(dev), k).cpu().long() == want).all(-1).float().mean().item()
    print(f"tie-rich lattice B{B} N{N} C{C} k{k}: rows equal to torch.topk's order {rows:.4f} (lowest-index rule: {plain:.4f})")
    assert rows == 1.0


def test_knn_exact_equals_default_without_ties(dev, ref):
    """on tie-free rows the two searches agree (and the exact one costs one more list entry plus a check per row)"""
    from hs_pose_amd import ops
    x = ref.hash_tensor((2, 1028, 128), 9100, 1.0).to(dev)
    with ops.exact_scope(True):
        a = ops.knn(x, 20)
    assert torch.equal(a, ops.knn(x, 20))
