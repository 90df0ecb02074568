"""GPU parity of the bf16 feature-storage path (BASELINE.json configs[3]: B=64 N=4096 bf16, "KNN LDS tiling + MFMA MLP
path"); the reference has no bf16 mode of its own -- its rule is only that the receptive field stays fp32
(network/fs_net_repo/gcn3d.py:57,59) -- so the yardstick is the fp32 path / fp32 oracle on the SAME bf16-rounded inputs.

Tolerance definition (stated once, used below):
  * a *_bf16 kernel computes exactly what its fp32 twin computes on the widened inputs and rounds the result once to bf16
    (round to nearest even).  Deterministic kernels (forward passes, BatchNorm, concat, CSR gather, weight gradient) must
    therefore equal  twin(widened inputs).bfloat16()  BIT FOR BIT; kernels that accumulate with LDS float atomics are held
    to one bf16 ulp (2^-8 relative) plus the twin's own order noise.
  * hsp_knn_bf16: bf16 x bf16 products are exact in fp32, the MFMA adds 16 of them per step in its own order, so indices
    need not equal an fp32-chain evaluation on near ties: the neighbour-SET agreement is reported and held above 97 % of
    rows on well-separated data.
  * one HS layer against the fp32 CPU oracle on the bf16-rounded inputs / weights at N = 4096: 1e-2 of scale (measured
    3.5e-3: fm, F and out are each rounded once).
  * whole stack against the fp32 path at B = 2, N = 4096 (and 1028), fp32 path's feature-space neighbour lists replayed
    (selection is discontinuous, tests/test_gpu_stack.py): feat max error <= 6e-2 of its scale, RMS error <= 8e-2 of its RMS;
    parameter gradients: norm within 25 %, cosine >= 0.88.  Measured (round 2): 3.3-4.1e-2 / 4.1-4.6e-2; gradient cosines
    0.91-0.999.  Why not 4e-3: every layer is within 2.5e-3 of its fp32 twin (tools/diag_bf16_layers.py), but with
    random-init weights a layer output's per-channel MEAN is ~10x its standard deviation, and train-mode BatchNorm keeps only
    the deviation -- so 0.25 % of the magnitude becomes ~3 % of what survives normalisation (tools/diag_bf16_stack.py:
    2e-3 before bn1, 3e-2 after); this is the price of bf16 feature storage itself (torch autocast pays it too), not of a
    kernel.  Free-running agreement is printed, not asserted."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
GRAD_NORM_TOL, GRAD_COS_TOL = 0.25, 0.88


def _h(ref, shape, seed, scale=1.0):
    return ref.hash_tensor(shape, seed, scale)


def test_cast_params_matches_rne(dev, ref):
    from hs_pose_amd.ops_bf16 import Bf16Params
    ws = [_h(ref, (128, 1024), 1).to(dev), _h(ref, (70, 33), 2).to(dev), _h(ref, (256, 512), 3).to(dev)[:, :256]]
    bp = Bf16Params([(ws[0], True, True), (ws[1], True, True), (ws[2], False, True)])
    bp.refresh()
    for (w, c, ct) in bp.entries:
        if c is not None:
            assert torch.equal(c, w.bfloat16())
        assert torch.equal(ct, w.t().contiguous().bfloat16())


@pytest.mark.parametrize("B,N,C,k", [(2, 1024, 128, 20), (2, 257, 256, 20), (1, 4096, 128, 20), (2, 100, 64, 8), (1, 64, 32, 2)])
def test_knn_bf16_vs_fp32_on_same_rows(dev, ref, B, N, C, k):
    from hs_pose_amd import ops
    x = torch.relu(_h(ref, (B, N, C), 11, 1.0)).to(dev).bfloat16()
    got = ops.knn(x, k).long()
    want = ops.knn(x.float(), k).long()                     # the bit-exact fp32 chain on the widened rows
    assert got.min() >= 0 and got.max() < N
    same_set = (got.sort(dim=2)[0] == want.sort(dim=2)[0]).all(dim=2).float().mean().item()
    same_order = (got == want).all(dim=2).float().mean().item()
    print(f"knn_bf16 B{B} N{N} C{C} k{k}: rows with the fp32 neighbour set {same_set:.4f}, same order {same_order:.4f}")
    assert same_set > 0.97
    # every picked neighbour is within rounding of the true k-th distance
    xf = x.float()
    d = torch.cdist(xf, xf) ** 2
    dk = torch.gather(d, 2, want)[:, :, -1:]
    dg = torch.gather(d, 2, got)
    assert (dg <= dk * (1 + 1e-3) + 1e-3).all()


@pytest.mark.parametrize("B,N,k,S,C", [(2, 257, 20, 7, 256), (2, 1028, 20, 7, 128), (1, 4096, 20, 7, 128), (2, 64, 8, 3, 32)])
def test_rf_conv_bf16_equals_rounded_fp32_twin(dev, ref, B, N, k, S, C):
    from hs_pose_amd import ops, ops_bf16
    from hs_pose_amd._lib import lib
    xyz = _h(ref, (B, N, 3), 21, 0.05).to(dev)
    dirs = _h(ref, (3, S * C), 22, 0.3).to(dev)
    fm = _h(ref, (B, N, (S + 1) * C), 23, 1.0).to(dev).bfloat16()
    feat = torch.relu(_h(ref, (B, N, 64), 24, 1.0)).to(dev)
    idx = ops.knn(feat, k)
    out_b, arg_b, fwin_b = ops_bf16._rf_conv_fwd(xyz, idx, dirs, fm, S, True)
    out_f, arg_f, fwin_f = ops._rf_conv_fwd_raw(xyz, idx, dirs, fm.float(), S, True)
    assert torch.equal(arg_b, arg_f)
    assert torch.equal(out_b, out_f.bfloat16())
    if fwin_b is not None and fwin_f is not None:
        assert torch.equal(fwin_b, fwin_f.bfloat16())
    g = _h(ref, (B, N, C), 25, 1.0).to(dev).bfloat16()
    saved_b = fwin_b if fwin_b is not None else fm
    saved_f = fwin_f if fwin_f is not None else fm.float()
    gfm_b, gd_b = ops_bf16._rf_conv_bwd(xyz, dirs, saved_b, arg_b, g, S)
    gfm_f, gd_f = ops._rf_conv_bwd_raw(xyz, idx, dirs, saved_f, arg_f, g.float(), S)
    scale = gfm_f.abs().max().item()
    err = (gfm_b.float() - gfm_f).abs()
    assert (err <= gfm_f.abs() * 2.0 ** -8 + 2e-5 * scale).all(), err.max().item() / scale
    assert (gd_b - gd_f).abs().max().item() <= 1e-4 * gd_f.abs().max().item()


def test_surface_gather_orl_bn_concat_bf16_twins(dev, ref):
    from hs_pose_amd import ops, ops_bf16
    B, N, k, S, C = 2, 1028, 20, 7, 128
    xyz = _h(ref, (B, N, 3), 31, 0.05).to(dev)
    idx = ops.knn(xyz, k)
    # neighbourhood max-pool (Pool_layer) forward: exact; backward: one bf16 rounding of the fp32 twin's sums
    f = _h(ref, (B, N, C), 32, 1.0).to(dev).bfloat16().requires_grad_(True)
    ff = f.detach().float().requires_grad_(True)
    sel = torch.randperm(N, generator=torch.Generator().manual_seed(1))[:N // 4].to(dev, torch.int32)
    pb, pf = ops.gather_max(f, idx, 4, qsel=sel), ops.gather_max(ff, idx, 4, qsel=sel)
    assert pb.dtype == BF and torch.equal(pb, pf.bfloat16())
    g = _h(ref, tuple(pb.shape), 33, 1.0).to(dev).bfloat16()
    pb.backward(g); pf.backward(g.float())
    assert ((f.grad.float() - ff.grad).abs() <= ff.grad.abs() * 2.0 ** -8 + 1e-5).all()
    # ORL global feature: fp32 (B,C) out, identical to the twin
    fg_b, arg_b = ops_bf16._orl_fwd(f.detach(), idx, k)
    fg_f, arg_f = ops._orl_fwd_raw(ff.detach(), idx, k)
    assert torch.equal(arg_b, arg_f) and torch.equal(fg_b, fg_f)
    assert torch.equal(ops_bf16._colsum(f.detach()), ops.colsum_rows(ff.detach()))
    # fused train-mode BatchNorm + ReLU: statistics in fp32 from the same values in the same order
    bn_b, bn_f = torch.nn.BatchNorm1d(C).to(dev), torch.nn.BatchNorm1d(C).to(dev)
    with torch.no_grad():
        bn_b.weight.copy_(_h(ref, (C,), 34, 1.0).to(dev)); bn_f.weight.copy_(bn_b.weight)
        bn_b.bias.copy_(_h(ref, (C,), 35, 0.5).to(dev)); bn_f.bias.copy_(bn_b.bias)
    xb = f.detach().clone().requires_grad_(True)
    xf = ff.detach().clone().requires_grad_(True)
    yb, yf = ops.bn_relu(xb, bn_b), ops.bn_relu(xf, bn_f)
    assert torch.equal(yb, yf.bfloat16())
    assert torch.equal(bn_b.running_mean, bn_f.running_mean) and torch.equal(bn_b.running_var, bn_f.running_var)
    gy = _h(ref, tuple(yb.shape), 36, 1.0).to(dev).bfloat16()
    yb.backward(gy); yf.backward(gy.float())
    assert torch.equal(xb.grad, xf.grad.bfloat16())
    assert torch.equal(bn_b.weight.grad, bn_f.weight.grad) and torch.equal(bn_b.bias.grad, bn_f.bias.grad)
    # eval-mode BatchNorm on bf16 rows
    bn_b.eval(); bn_f.eval()
    assert torch.allclose(ops.bn_relu(xb.detach(), bn_b).float(), ops.bn_relu(xf.detach(), bn_f), atol=2e-2, rtol=2 ** -7)
    # surface layer's graph conv: bf16 out of fp32 geometry
    dirs = _h(ref, (3, S * C), 37, 0.3).to(dev)
    F3 = torch.empty(B, N, C, dtype=BF, device=dev)
    arg = torch.empty(B, N, S * C, dtype=torch.uint16, device=dev)
    ops._run("hsp_rf_surface_fwd_bf16", (ops._p(xyz), ops._p(idx), ops._p(dirs), B, N, k, S, C, ops._p(F3), ops._p(arg), ops._stream()))
    twin = ops.rf_surface(xyz, idx, dirs, S)
    assert torch.equal(F3, twin.bfloat16())


def test_assemble_feat_and_upsample_backward_bf16(dev, ref):
    from hs_pose_amd import ops
    B, N, N1 = 2, 1028, 257
    a = _h(ref, (B, N, 128), 41, 1.0).to(dev).bfloat16().requires_grad_(True)
    c = _h(ref, (B, N1, 256), 42, 1.0).to(dev).bfloat16().requires_grad_(True)
    one_hot = torch.zeros(B, 6, device=dev); one_hot[0, 2] = 1; one_hot[1, 5] = 1
    near = torch.randint(0, N1, (B, N), generator=torch.Generator().manual_seed(2)).to(dev, torch.int32)
    feat = ops.assemble_feat([(a, None, 0), (c, near, 1), (one_hot, None, 2)])
    assert feat.dtype == BF and feat.shape == (B, N, 390)
    want = torch.cat([a, torch.gather(c, 1, near.long().unsqueeze(-1).expand(-1, -1, 256)),
                      one_hot.bfloat16().unsqueeze(1).expand(-1, N, -1)], dim=2)
    assert torch.equal(feat, want)
    g = _h(ref, (B, N, 390), 43, 1.0).to(dev).bfloat16()
    feat.backward(g)
    assert torch.equal(a.grad, g[:, :, :128])
    ref_c = torch.zeros(B, N1, 256, device=dev).index_put_((torch.arange(B, device=dev).unsqueeze(1).expand(-1, N), near.long()),
                                                          g[:, :, 128:384].float(), accumulate=True)
    assert ((c.grad.float() - ref_c).abs() <= ref_c.abs() * 2.0 ** -8 + 1e-5).all()


@pytest.mark.parametrize("K,M,N,colsum,pitch", [(16448, 128, 1024, True, 0), (4112, 256, 2048, False, 0), (1000, 128, 128, True, 0),
                                                 (65536, 128, 1024, True, 0), (5000, 128, 256, True, 8), (300, 64, 192, True, 0)])
def test_wgrad_bf16(dev, ref, K, M, N, colsum, pitch):
    """A^T B (+ column sums of B) of bf16 point rows: exact bf16 x bf16 products accumulated in fp32 -- on the bf16 matrix cores
    when M, N are multiples of 128 (8 x 8 register transposes into the LDS operand image), else by widening onto the fp32
    ones.  Against float64, within fp32 accumulation error of the summed magnitudes; `pitch`: rows are slices of wider tensors."""
    from hs_pose_amd import ops_bf16
    A = _h(ref, (K, M + pitch), 51, 1.0).to(dev).bfloat16()[:, :M]
    Bm = _h(ref, (K, N + pitch), 52, 1.0).to(dev).bfloat16()[:, pitch:]
    out = ops_bf16._wgrad(A, Bm, colsum=colsum)
    gw, cs = out if colsum else (out, None)
    want = A.double().t() @ Bm.double()
    mag = A.double().abs().t() @ Bm.double().abs()
    assert ((gw.double() - want).abs() <= 1e-6 * mag + 1e-6).all()
    if colsum:
        wc, mc = Bm.double().sum(0), Bm.double().abs().sum(0)
        assert ((cs.double() - wc).abs() <= 1e-6 * mc + 1e-6).all()
    again = ops_bf16._wgrad(A, Bm, colsum=colsum)
    assert torch.equal(gw, again[0] if colsum else again)          # fixed summation order


def _twin_nets(ref, flags, dev, seed=0):
    from hs_pose_amd.FaceRecon import FaceRecon
    flags.train = 0
    nets = []
    for dt in (torch.float32, BF):
        torch.manual_seed(seed)
        net = FaceRecon().to(dev).train()
        net.set_feature_dtype(dt)
        nets.append(net)
    return nets


class _Replay:
    """record the fp32 path's feature-space neighbour lists, replay them in the bf16 path (tests/test_gpu_stack.py)"""

    def __init__(self, monkeypatch):
        from hs_pose_amd import ops
        self.real, self.lists, self.mode, self.pos, self.agree = ops.knn, [], "record", 0, []
        monkeypatch.setattr(ops, "knn", self)

    def __call__(self, x, k, drop_first=True):
        own = self.real(x, k, drop_first)
        if x.shape[-1] == 3:
            return own
        if self.mode == "record":
            self.lists.append(own)
            return own
        want = self.lists[self.pos]; self.pos += 1
        self.agree.append((own.sort(dim=2)[0] == want.sort(dim=2)[0]).all(dim=2).float().mean().item())
        return want if self.mode == "replay" else own


@pytest.mark.parametrize("B,N", [(2, 1028), (2, 4096)])
def test_hs_stack_bf16_vs_fp32_path(dev, ref, flags, monkeypatch, B, N):
    """the whole HS stack, forward + backward, bf16 storage against the fp32 path (itself pinned to the reference) on the
    same cloud, weights and pool draws -- at the dense-cloud size of configs[3] (N = 4096) and at N = 1028."""
    net_f, net_b = _twin_nets(ref, flags, dev)
    pts = _h(ref, (B, N, 3), 61, 0.05).to(dev)
    pts = pts - pts.mean(dim=1, keepdim=True)
    obj = torch.tensor([[1.0], [4.0]])[:B].to(dev)
    dfeat = _h(ref, (B, N, 1286), 62, 1.0).to(dev)
    rp = _Replay(monkeypatch)
    torch.manual_seed(5)
    _, _, feat_f = net_f(pts, obj)
    feat_f.backward(dfeat)
    rp.mode = "replay"
    torch.manual_seed(5)
    _, _, feat_b = net_b(pts, obj)
    assert feat_b.dtype == BF
    feat_b.backward(dfeat.bfloat16())
    scale, rms = feat_f.abs().max().item(), feat_f.pow(2).mean().sqrt().item()
    err = (feat_b.float() - feat_f)
    emax, erms = err.abs().max().item() / scale, err.pow(2).mean().sqrt().item() / rms
    worst, table = 0.0, []
    for (n_, pf), (_, pb) in zip(net_f.named_parameters(), net_b.named_parameters()):
        nf, nb = pf.grad.norm().item(), pb.grad.norm().item()
        cos = torch.nn.functional.cosine_similarity(pf.grad.flatten(), pb.grad.flatten(), dim=0).item()
        table.append((n_, abs(nb - nf) / max(nf, 1e-12), cos))
        worst = max(worst, abs(nb - nf) / max(nf, 1e-12))
    print("  parameter gradients, bf16 path vs fp32 path (norm deviation, cosine): " +
          "; ".join(f"{n_} {d:.1e} {c:.4f}" for n_, d, c in table))
    print(f"BF16 STACK B{B} N{N}: own feature-KNN rows agreeing with the fp32 path's sets per layer {[round(a, 4) for a in rp.agree]}; "
          f"feat max err {emax:.2e} of scale, rms err {erms:.2e} of rms; worst parameter-gradient norm deviation {worst:.2e}")
    assert emax <= 6e-2 and erms <= 8e-2                # see the module docstring
    assert worst <= GRAD_NORM_TOL and min(c for _, _, c in table) >= GRAD_COS_TOL, (worst, min(c for _, _, c in table))
    # free-running (no replay): agreement and error are REPORTED (selection is discontinuous), only sanity-asserted
    rp.mode, rp.pos, rp.agree = "watch", 0, []
    torch.manual_seed(5)
    _, _, feat_w = net_b(pts, obj)
    e2 = (feat_w.float() - feat_f).abs().max().item() / scale
    print(f"BF16 STACK B{B} N{N} free-running: set agreement per layer {[round(a, 4) for a in rp.agree]}; feat max err {e2:.2e} of scale")
    assert torch.isfinite(feat_w.float()).all()
    # measured (round 3), rows whose free-running neighbour SET equals the fp32 path's: 0.52 / 0.33 / 0.19 / 0.58 at N = 1028,
    # 0.37 / 0.18 / 0.10 / 0.13 at N = 4096, feat max error 0.17-0.18 of scale: a bf16 ulp (4e-3) against near-tied expanded
    # distances re-routes most rows' last neighbours, and every later layer inherits the earlier ones' re-routing.  Floors at
    # about half the measured figures.
    assert min(rp.agree) >= 0.05 and rp.agree[0] >= 0.2, rp.agree
    assert e2 <= 0.5, e2


@pytest.mark.parametrize("lname,N,Cin,k", [("conv_1", 4096, 128, 20), ("conv_2", 1024, 128, 20), ("conv_3", 1024, 256, 20),
                                           ("conv_4", 256, 256, 20), ("conv_1", 1028, 128, 20)])
def test_hs_layer_bf16_vs_cpu_oracle(dev, ref, flags, lname, N, Cin, k):
    """every HS layer of the stack at the cloud sizes of BASELINE configs[3] (N = 4096 -> 1024 -> 256) against the fp32 CPU
    ORACLE (oracle/ref_cpu.py) evaluated on the bf16-rounded input and weights, with the oracle's own neighbour list:
    forward 1e-2 of scale (the layer rounds fm, F and out to bf16)."""
    from hs_pose_amd import gcn3d, ops, ops_bf16
    from hs_pose_amd.FaceRecon import FaceRecon
    flags.train = 0
    torch.manual_seed(0)
    net = FaceRecon().to(dev)
    net.set_feature_dtype(BF)
    layer = getattr(net, lname)
    B = 1
    xyz = _h(ref, (B, N, 3), 71, 0.05)
    X = torch.relu(_h(ref, (B, N, Cin), 72, 1.0)).bfloat16()
    p = {f"{lname}.{n_}": v.detach().cpu().bfloat16().float() if n_ in ("weights", "STE_layer.weight", "conv2.weight") else v.detach().cpu()
         for n_, v in layer.state_dict().items()}
    want = ref.hs_layer(p, lname + ".", xyz, X.float(), k, 7)
    idx_f = ref.knn_index(X.float(), k).to(torch.int32).to(dev)
    net._bf16.refresh()
    with gcn3d.knn_scope():
        got = ops_bf16.hs_layer(
            xyz.to(dev), X.to(dev), idx_f, ops.knn(xyz.to(dev), k), k, 7, layer.weights, layer.bias, layer.directions,
            layer.STE_layer.weight, layer.conv2.weight)
    err = (got.float().cpu() - want).abs().max().item() / want.abs().max().item()
    print(f"bf16 {lname} vs fp32 CPU oracle (N={N}): max err {err:.2e} of scale")
    assert err <= 1e-2


def test_bf16_config3_full_batch(dev, ref, flags):
    """BASELINE configs[3] at its stated size, B = 64, N = 4096, bf16 feature storage: the hipGraph replay bench.py times equals
    the eager step (feat and every parameter gradient), everything is finite, and the feature-space neighbour lists the
    step used have the KNN properties (k distinct in-range rows per query; the list of a query equals the C-ABI kernel's on
    the widened rows = the fp32 chain, on a sample of clouds)."""
    from hs_pose_amd import gcn3d, ops
    from hs_pose_amd.FaceRecon import FaceRecon
    from hs_pose_amd.graph import GraphedStep
    B, N = 64, 4096
    g = torch.Generator().manual_seed(9)
    pc = torch.randn(B, N, 3, generator=g) * 0.05
    centred = (pc - pc.mean(dim=1, keepdim=True)).to(dev)
    obj = torch.randint(0, 6, (B, 1), generator=g).float().to(dev)
    dfeat = torch.randn(B, N, 1286, generator=g).bfloat16().to(dev)

    def make():
        flags.train = 0
        torch.manual_seed(0)
        return FaceRecon().to(dev).train().set_feature_dtype(BF)
    net_g, net_e = make(), make()
    torch.manual_seed(11)
    graphed = GraphedStep(net_g, centred, obj, dfeat, warmup=1)
    for _ in range(2):
        feat_g = graphed.run()
    torch.cuda.synchronize()
    seen = []
    real = ops.knn

    def spy(x, k, drop_first=True):
        out = real(x, k, drop_first)
        if x.shape[-1] != 3:
            seen.append((x, k, out))
        return out
    ops.knn = spy
    try:
        with gcn3d.pool_index_feed([p.clone() for p in graphed.pool_idx]):
            _, _, feat_e = net_e(centred, obj)
    finally:
        ops.knn = real
    feat_e.backward(dfeat)
    torch.cuda.synchronize()
    assert feat_g.dtype == BF and feat_g.shape == (B, N, 1286)
    assert torch.isfinite(feat_g.float()).all() and torch.isfinite(feat_e.float()).all()
    scale = feat_e.float().abs().max().item()
    assert (feat_g.float() - feat_e.float()).abs().max().item() <= 1e-5 * scale        # same kernels, same order: equal
    gmax = max(p.grad.abs().max().item() for p in net_e.parameters() if p.grad is not None)
    for (k_, pg), (_, pe) in zip(net_g.named_parameters(), net_e.named_parameters()):
        if pe.grad is None:
            continue
        assert torch.isfinite(pg.grad).all(), k_
        # (the LDS-atomic backward kernels sum in arrival order: one bf16 ulp of a summand per element)
        assert (pg.grad - pe.grad).abs().max().item() <= 2e-3 * gmax, k_
    assert len(seen) == 4
    for x, k, idx in seen:                                   # (B, n, C) bf16 rows, (B, n, k) lists
        n = x.shape[1]
        assert idx.shape == (B, n, k) and int(idx.min()) >= 0 and int(idx.max()) < n
        srt = idx.sort(dim=2)[0]
        assert bool((srt[:, :, 1:] != srt[:, :, :-1]).all()), "a neighbour list repeats a row"
        wide = ops.knn(x[:2].float().contiguous(), k)        # the fp32 chain on the widened rows
        same = (wide.sort(dim=2)[0] == idx[:2].sort(dim=2)[0]).all(dim=2).float().mean().item()
        assert same > 0.97, same
