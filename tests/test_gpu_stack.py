"""GPU parity of the whole HS stack / PoseNet9D against the reference's golden outputs (pose/size
outputs within 1e-4, BASELINE north_star) and of the backward of unit U1 (feat -> all HS parameters)."""
import os

import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu

# Free-running (no teacher forcing) bounds at N=1028.  Yardstick: the CPU oracle's OWN drift when its input moves by
# 1 ulp (tools/oracle_free_running_sensitivity.py, 3 noise seeds): eval-mode BN -> rows with identical neighbour sets per
# HS layer 0.87 / 0.60-0.67 / 0.33-0.40 / 0.27-0.41, pose / size outputs move by up to 7.9e-4; train-mode BN (one
# re-routed row reaches every row) -> 0.86 / 0.62-0.65 / 0.39-0.41 / 0.66-0.70 and up to 9.8e-2.  Measured on the GPU
# (round 2): 0.92 / 0.69 / 0.43 / 0.41 with 8.8e-4, and 0.92 / 0.83 / 0.64 / 0.82 with 6.3e-2 -- inside the reference's
# own conditioning.  Bounds = ~3x / ~2x the oracle's worst drift; agreement floors below every oracle-vs-oracle figure.
FREE_RUNNING_BOUND = {"stack_eval_1028": 3e-3, "stack_evalflags_trainbn_1028": 2e-1}
FREE_RUNNING_AGREE = {"stack_eval_1028": (0.8, 0.5, 0.25, 0.2), "stack_evalflags_trainbn_1028": (0.8, 0.5, 0.25, 0.4)}

OUT_NAMES = ["recon", "face_normal", "face_dis", "face_f", "p_green_R", "p_red_R", "f_green_R", "f_red_R", "Pred_T", "Pred_s"]


def _build(ref, flags, dev, train_flag, bn_training):
    from hs_pose_amd.PoseNet9D import PoseNet9D
    flags.train = train_flag
    net = PoseNet9D()
    sd = net.state_dict()
    ref.fill_state_closed_form(sd)
    net = net.to(dev)
    net.train(bn_training)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return net


def _inputs(ref, B, N, seed, dev):
    pts = ref.hash_tensor((B, N, 3), seed, 0.05)
    pts[:, :, 2] += 0.8
    obj = torch.from_numpy((ref.hash_unit(B, seed + 1) * 6).astype(np.int64)).float().view(B, 1)
    return pts.to(dev), obj.to(dev)


def _maxerr(a, b):
    return (a.detach().cpu().double() - torch.from_numpy(np.asarray(b)).double()).abs().max().item()


class ForcedFeatKnn:
    """Teacher-forcing of the feature-space neighbour sets (DESIGN.md "selection discontinuity").

    KNN selection is discontinuous: the reference ranks post-ReLU features by the cancellation-prone
    expanded distance, so 1e-7 rounding differences upstream (GEMM / BN summation order) flip a few
    near-tied neighbours per thousand rows, and train-mode BatchNorm then spreads such a flip to every
    row.  Bit-exact index parity is therefore asserted where it is well defined -- on IDENTICAL inputs
    (tests/test_gpu_knn.py, test_hs_layer_golden) -- and the stack is compared with the reference at
    1e-4 under the reference's own neighbour sets, replayed here in call order.  The kernel still runs
    on the GPU's own features; its agreement with the replayed sets is recorded in .agree."""

    def __init__(self, monkeypatch, g, dev, force=True):
        from hs_pose_amd import ops
        self.force = force
        self.real = ops.knn
        self.lists = [torch.from_numpy(g[f"featknn{i}"].astype(np.int32)).to(dev) for i in (1, 2, 3, 4)]
        self.calls = 0
        self.agree = []                    # rows whose ordered neighbour list equals the reference's
        self.agree_set = []                # rows whose neighbour SET does
        monkeypatch.setattr(ops, "knn", self)

    def __call__(self, x, k, drop_first=True, **kw):
        own = self.real(x, k, drop_first, **kw)
        if x.shape[-1] == 3:
            return own
        want = self.lists[self.calls % 4]
        self.calls += 1
        assert want.shape == own.shape
        self.agree.append((own == want).all(dim=2).float().mean().item())
        self.agree_set.append((torch.sort(own, dim=2)[0] == torch.sort(want, dim=2)[0]).all(dim=2).float().mean().item())
        return want if self.force else own


@pytest.mark.parametrize("name", ["stack_eval_256", "stack_eval_1028", "stack_evalflags_trainbn_1028", "stack_train_256"])
def test_posenet9d_golden(dev, ref, flags, monkeypatch, gemm_mode, name):
    """pose / size outputs of PoseNet9D within 1e-4 of the reference (BASELINE north_star)."""
    g = golden(name)
    train_flag, B, N, seed, bn_training = (int(v) for v in g["meta"])
    net = _build(ref, flags, dev, train_flag, bool(bn_training))
    pts, obj = _inputs(ref, B, N, seed, dev)
    forced = ForcedFeatKnn(monkeypatch, g, dev)
    torch.manual_seed(1)                       # the two Pool_layer randperm draws
    outs = dict(zip(OUT_NAMES, net(pts, obj)))
    assert forced.calls == 4
    print(f"{name}: rows whose own feature-KNN equals the reference's, per HS layer: {forced.agree}")
    assert min(forced.agree) > 0.9
    for n_ in OUT_NAMES[4:]:
        err = _maxerr(outs[n_], g["out." + n_])
        assert err <= 1e-4, f"{name} {n_}: {err:.3e}"
    if train_flag:
        for n_ in OUT_NAMES[:4]:
            err = _maxerr(outs[n_].reshape(-1)[::211], g["outsample." + n_])
            assert err <= 1e-4 * max(1.0, np.abs(g["outsample." + n_]).max()), f"{name} {n_}: {err:.3e}"
    else:
        assert all(outs[n_] is None for n_ in OUT_NAMES[:4])
    torch.manual_seed(1)
    _, _, feat = net.face_recon(torch.from_numpy(g["centred"]).to(dev), obj)
    assert feat.shape == (B, N, 1286)
    scale = max(1.0, np.abs(g["feat_sample"]).max())
    assert _maxerr(feat.reshape(-1)[::1009], g["feat_sample"]) <= 1e-4 * scale
    assert _maxerr(feat.mean(dim=(0, 1)), g["feat_chmean"]) <= 1e-4 * scale


def test_posenet9d_free_running_eval_256(dev, ref, flags, monkeypatch):
    """the same comparison WITHOUT teacher forcing on the small eval-mode case.  If this run's own
    feature-space neighbour sets equal the reference's everywhere, the outputs must meet 1e-4; if a
    near-tie flipped somewhere (rounding-order dependent, see ForcedFeatKnn) only a loose bound holds."""
    g = golden("stack_eval_256")
    train_flag, B, N, seed, bn_training = (int(v) for v in g["meta"])
    net = _build(ref, flags, dev, train_flag, bool(bn_training))
    pts, obj = _inputs(ref, B, N, seed, dev)
    watch = ForcedFeatKnn(monkeypatch, g, dev, force=False)
    torch.manual_seed(1)
    outs = dict(zip(OUT_NAMES, net(pts, obj)))
    exact = min(watch.agree) == 1.0
    print(f"free-running: own feature-KNN == reference per layer: {watch.agree}")
    assert min(watch.agree) > 0.8
    for n_ in OUT_NAMES[4:]:
        assert _maxerr(outs[n_], g["out." + n_]) <= (1e-4 if exact else 2e-2), n_


@pytest.mark.parametrize("name,free_exact", [("stack_evalflags_trainbn_1028", False), ("stack_train_256", False),
                                             ("stack_evalflags_trainbn_1028", True), ("stack_train_256", True)])
def test_hs_stack_backward_golden(dev, ref, flags, monkeypatch, gemm_mode, name, free_exact):
    """unit U1: feat from the centred cloud (train-mode BN), backward from a closed-form dfeat to every
    HS-stack parameter; also BN running statistics after the step.  ``free_exact``: the same bounds WITHOUT replaying the
    reference's feature-space lists -- the forward in the reference's rounding order (FaceRecon.exact_train) finds them itself."""
    from hs_pose_amd.FaceRecon import FaceRecon
    g = golden(name)
    train_flag, B, N, seed, bn_training = (int(v) for v in g["meta"])
    net = _build(ref, flags, dev, train_flag, True)
    _, obj = _inputs(ref, B, N, seed, dev)
    if free_exact:
        monkeypatch.setattr(FaceRecon, "exact_train", True)
    watch = ForcedFeatKnn(monkeypatch, g, dev, force=not free_exact)
    torch.manual_seed(1)
    _, _, feat = net.face_recon(torch.from_numpy(g["centred"]).to(dev), obj)
    if free_exact:
        print(f"{name} free-running, reference-order arithmetic: own lists == reference's on {[round(a, 4) for a in watch.agree]} of the rows")
        assert all(a >= 0.999 for a in watch.agree), watch.agree
    dfeat = ref.hash_tensor(tuple(feat.shape), seed + 5, 1.0).to(dev)
    (feat * dfeat).sum().backward()
    checked = 0
    for pn, p in net.face_recon.named_parameters():
        key = "gradsample." + pn
        if key not in g.files:
            assert p.grad is None or pn.split(".")[0] in ("conv1d_block", "recon_head", "face_head"), pn
            continue
        want = g[key]
        norm, _ = g["gradnorm." + pn]
        got = p.grad.reshape(-1)[::499].cpu().double().numpy()
        # Tolerance: the stack is only piecewise smooth (ReLU masks, arg-max routes, xyz-KNN sets), and a
        # 1e-7 forward difference toggles a few of ~10^6 kinks, each moving one summand of a parameter
        # gradient.  tools/oracle_sensitivity.py measures the CPU oracle's OWN gradient drift under 1-ulp
        # input noise at 1e-3..7e-3 of the norm; the bound below is inside that.  Strict 1e-4 gradient
        # parity is asserted per layer on identical inputs (tests/test_gpu_layers.py), and the last
        # layer's gradient norms (conv_4: fewest kinks upstream) are held to 1e-4 here.
        last = pn.startswith("conv_4.")
        tol = 3e-2 * max(np.abs(want).max(), norm / max(p.numel(), 1) ** 0.5, 1e-12)
        assert np.abs(got - want).max() <= tol, f"{name} {pn}: {np.abs(got - want).max():.3e} > {tol:.3e}"
        gn = p.grad.double().norm().item()
        assert abs(gn - norm) <= (1e-4 if last else 3e-3) * max(norm, 1e-12), f"{name} {pn}: grad norm {gn} vs {norm}"
        checked += 1
    assert checked >= 26
    for bn_ in ("bn1", "bn2", "bn3"):
        for st in ("running_mean", "running_var"):
            want = g[f"bnstat.{bn_}.{st}"]
            got = getattr(getattr(net.face_recon, bn_), st).cpu().numpy()
            assert np.abs(got - want).max() <= 1e-4 * max(1.0, np.abs(want).max()), (bn_, st)


def test_state_dict_roundtrip(dev, ref, flags, state_keys):
    """a reference-shaped checkpoint loads strict=True, incl. the eval-side key filtering/renaming of
    evaluation/evaluate.py:62-73 (drop train heads, resconv -> STE_layer)."""
    from hs_pose_amd.PoseNet9D import PoseNet9D
    flags.train = 1
    ck = {k_: torch.zeros(shape) for k_, shape in state_keys["train"].items()}
    flags.train = 0
    net = PoseNet9D()
    drop = ("face_recon.conv1d_block", "face_recon.face_head", "face_recon.recon_head")
    sd = {k_.replace("resconv", "STE_layer"): v for k_, v in ck.items() if not k_.startswith(drop)}
    net.load_state_dict(sd, strict=True)
    assert {k_: list(v.shape) for k_, v in net.state_dict().items()} == state_keys["eval"]


def test_cpu_input_fails_loudly(ref, flags):
    """no eager / CPU fallback: a CPU tensor is an error, not a silent slow path."""
    from hs_pose_amd import gcn3d
    from hs_pose_amd._lib import HspError
    with pytest.raises(HspError):
        gcn3d.get_neighbor_index(torch.zeros(1, 32, 3), 4)


@pytest.mark.parametrize("split", ["0", "1", "probe"])
def test_bench_data_parallel_path_smoke(dev, split):
    """bench.py with a forced 1-rank RCCL process group: hipGraph replay + the gradient exchange per step (the N>1 code
    path, minus the peers) -- one flat-buffer all-reduce, and the two-graph form with the first all-reduce issued
    asynchronously under the second graph (what more than one rank runs)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSP_FORCE_DIST="1", HSP_SPLIT_GRAPH=split, MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29533 + {"0": 0, "1": 1, "probe": 2}[split]), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "2", "--batch", "2",
                          "--points", "256", "--no-cpu-baseline", "--no-gemm-tuning"], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["hipgraph"] is True and line["value"] > 0 and line["n_gpus"] == 1
    assert "capture (split=True) failed" not in out.stderr
    if split == "probe":           # both forms captured, the start-up probe's collectives (barrier, object gather / broadcast) over RCCL
        assert set(line["config"]["grad_exchange_choice"]["probe_ms_per_step"]) == {"split", "single"}
        assert line["config"]["process_group"]["backend"] == "nccl"
    else:
        assert line["config"]["split_graph"] is (split == "1")
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs on the node (RCCL over xGMI)")
def test_rccl_two_ranks(dev):
    """ready for the first multi-GPU lease: 2 RCCL ranks under torchrun -- (a) tests/_rccl_two_rank_check.py: after the
    graphed step's exchange ``flat_grad`` == mean of the per-rank gradients, both exchange forms; (b) bench.py --gpus 2
    prints one line with n_gpus == 2 and weak-scaled work."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    run = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1"]
    out = subprocess.run(run + ["--master-port", "29561", os.path.join(root, "tests", "_rccl_two_rank_check.py")], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "RCCL_TWO_RANK_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
    print(out.stdout)
    out = subprocess.run(run + ["--master-port", "29562", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup",
                                "2", "--batch", "4", "--points", "256", "--no-cpu-baseline", "--no-gemm-tuning"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 8 and line["scaling"] == "weak" and line["value"] > 0


@pytest.mark.parametrize("name", ["stack_eval_1028", "stack_evalflags_trainbn_1028"])
def test_posenet9d_free_running_1028(dev, ref, flags, monkeypatch, name):
    """What the product returns WITHOUT teacher forcing at the config-2 cloud size (N=1028): the network's own
    feature-space neighbour sets are used; agreement with the reference's recorded sets and the pose / size errors are
    reported (DESIGN.md section 2.2 quotes them) and held to a stated bound.  Where every neighbour row agrees the strict
    1e-4 applies; otherwise the outputs differ by what a few re-routed neighbours (and, under train-mode BatchNorm over
    B*N rows, their spread to every row) are worth -- the reference moves by the same order under 1-ulp input noise
    (tools/oracle_sensitivity.py)."""
    g = golden(name)
    train_flag, B, N, seed, bn_training = (int(v) for v in g["meta"])
    net = _build(ref, flags, dev, train_flag, bool(bn_training))
    pts, obj = _inputs(ref, B, N, seed, dev)
    watch = ForcedFeatKnn(monkeypatch, g, dev, force=False)
    torch.manual_seed(1)
    outs = dict(zip(OUT_NAMES, net(pts, obj)))
    errs = {n_: _maxerr(outs[n_], g["out." + n_]) for n_ in OUT_NAMES[4:]}
    exact = min(watch.agree) == 1.0
    print(f"FREE-RUNNING {name}: rows with the reference's neighbour set per HS layer {[round(a, 4) for a in watch.agree]}; "
          f"max abs error {({k_: float(f'{v:.2e}') for k_, v in errs.items()})}")
    report = os.environ.get("HSP_REPORT_DIR")
    if report:
        import json
        with open(os.path.join(report, f"free_running_{name}.json"), "w") as f:
            json.dump({"agree_rows_per_layer": watch.agree, "max_abs_err": errs}, f, indent=1)
    assert all(a >= f for a, f in zip(watch.agree, FREE_RUNNING_AGREE[name])), watch.agree
    bound = 1e-4 if exact else FREE_RUNNING_BOUND[name]
    for n_, e in errs.items():
        assert e <= bound, f"{name} {n_}: {e:.3e} > {bound}"


# Reference-INITIALISED weights (the parameters the reference itself starts from, torch.manual_seed(0)) instead of the closed-form
# fills: oracle/gen_golden_refinit.py.  The fixture carries the reference's own behaviour under a 1-ulp move of the cloud
# (`self_agree`, `self_drift`): rows with an identical neighbour set per HS layer 0.98 / 0.89-0.91 / 0.80-0.83 / 0.78-0.81 and
# pose / size drift 5e-4 ... 1.8e-3 in eval mode; 0.98 / 0.93-0.97 / 0.85-0.95 / 0.93-0.98 and 4e-2 ... 1.1e-1 under train-mode
# BatchNorm.  So the north star's 1e-4 does not hold free-running for the reference against itself either; the product is held to
# ~2x the reference's worst self-drift and to agreement floors below its self-agreement.
# Measured on the GPU (round 3): ordered lists identical on 0.90 / 0.74 / 0.58 / 0.62 of the rows and 2.8e-4 on p_green_R
# (Pred_T 1.1e-5, Pred_s 1.6e-5) in eval mode; 0.91 / 0.84 / 0.69 / 0.84 and 4.9e-2 under train-mode BatchNorm -- both INSIDE the
# reference's own 1-ulp drift, neither at 1e-4.  Those figures are with the dense products on the BLAS library; on the
# hand-written kernels (HSP_GEMM=own, the default: csrc/gemm_x3.hip sums each product row in another order) 0.91 / 0.70 / 0.54 /
# 0.57 and 1.3e-3 (p_red_R) in eval mode, 0.91 / 0.74 / 0.57 / 0.80 and 7.0e-2 under train-mode BatchNorm -- neighbour SETS
# 0.99 / 0.94 / 0.87-0.89 / 0.83-0.95, above the reference's self-agreement in every layer.  The test runs in both modes.
# Round 4: in EVAL mode the forward now runs in the reference's own rounding order (ops.exact_scope: k-ordered products, ATen's
# summation orders, torch.topk's tie order -- tests/test_gpu_exact.py pins each statement): every ordered neighbour list equals
# the reference's and the pose / size outputs agree to 2e-6 (1.3e-3 in round 3).  The train-mode case keeps the bounds above:
# BatchNorm batch statistics are a reduction over B*N rows whose order is the library's, not ATen's.
REFINIT_BOUND = {"stack_refinit_eval_1028": 1e-5, "stack_refinit_trainbn_1028": 1.5e-1}
REFINIT_AGREE = {"stack_refinit_eval_1028": (1.0, 1.0, 1.0, 1.0), "stack_refinit_trainbn_1028": (0.85, 0.65, 0.5, 0.7)}


def test_posenet9d_free_running_trainbn_exact_arithmetic(dev, ref, flags, monkeypatch):
    """TRAIN-mode BatchNorm, free-running, at the north star's 1e-4: with ``FaceRecon.exact_train`` the forward runs in the
    reference's own rounding order (the eval-mode arithmetic of round 4; the batch statistics stay this library's reduction) and
    the network's OWN feature-space searches return the reference's ordered lists on every row of every HS layer -- the fixture is
    the one the fast products meet only at 1.5e-1 (test below).  Forward with autograd recording, and a backward pass through it
    (the switch costs 24 % of the training step: off by default)."""
    from hs_pose_amd.FaceRecon import FaceRecon
    from hs_pose_amd.PoseNet9D import PoseNet9D
    g = golden("stack_refinit_trainbn_1028")
    _, B, N, seed, bn_training, wseed = (int(v) for v in g["meta"])
    assert bn_training == 1
    flags.train = 0
    monkeypatch.setattr(FaceRecon, "exact_train", True)
    torch.manual_seed(wseed)
    net = PoseNet9D().to(dev)
    net.train(True)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    pts, obj = _inputs(ref, B, N, seed, dev)
    watch = ForcedFeatKnn(monkeypatch, g, dev, force=False)
    torch.manual_seed(1)
    outs = dict(zip(OUT_NAMES, net(pts, obj)))                    # grad mode ON: the training forward itself
    errs = {n_: _maxerr(outs[n_], g["out." + n_]) for n_ in OUT_NAMES[4:]}
    print(f"FREE-RUNNING train-mode BatchNorm, reference-order arithmetic: ordered lists {[round(a, 4) for a in watch.agree]}, max abs error "
          f"{({k_: float(f'{v:.2e}') for k_, v in errs.items()})}")
    assert all(a >= 0.999 for a in watch.agree), watch.agree
    for n_, e in errs.items():
        assert e <= 1e-4, f"{n_}: {e:.3e}"
    sum(outs[n_].sum() for n_ in OUT_NAMES[4:]).backward()       # the backward runs through the exact forward's saved tensors
    gsq = sum(float(p.grad.double().pow(2).sum()) for p in net.parameters() if p.grad is not None)
    assert np.isfinite(gsq) and gsq > 0


@pytest.mark.parametrize("name", ["stack_refinit_eval_1028", "stack_refinit_trainbn_1028"])
def test_posenet9d_free_running_refinit(dev, ref, flags, monkeypatch, gemm_mode, name):
    """free-running PoseNet9D on reference-initialised weights at N = 1028: (1) the mirrored modules constructed under the
    reference's seed hold IDENTICAL parameters (samples + sums of every state tensor); (2) the network's own feature-space
    neighbour sets against the reference's, and the six pose / size outputs, inside the bounds above."""
    from hs_pose_amd.PoseNet9D import PoseNet9D
    g = golden(name)
    _, B, N, seed, bn_training, wseed = (int(v) for v in g["meta"])
    flags.train = 0
    torch.manual_seed(wseed)
    net = PoseNet9D()
    for k_, v in net.state_dict().items():
        if v.is_floating_point():
            flat = v.reshape(-1)
            assert np.array_equal(flat[::997].numpy(), g["wsample." + k_]), f"parameter {k_} differs from the reference's draw"
            assert abs(flat.double().sum().item() - g["wsum." + k_][0]) <= 1e-9 * max(1.0, g["wsum." + k_][1]), k_
    net = net.to(dev)
    net.train(bool(bn_training))
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    pts, obj = _inputs(ref, B, N, seed, dev)
    watch = ForcedFeatKnn(monkeypatch, g, dev, force=False)
    torch.manual_seed(1)
    with torch.no_grad():
        outs = dict(zip(OUT_NAMES, net(pts, obj)))
    errs = {n_: _maxerr(outs[n_], g["out." + n_]) for n_ in OUT_NAMES[4:]}
    print(f"FREE-RUNNING {name}: rows with the reference's ordered neighbour list per HS layer {[round(a, 4) for a in watch.agree]}, "
          f"with its neighbour set {[round(a, 4) for a in watch.agree_set]} "
          f"(sets, reference vs itself + 1 ulp: {np.round(g['self_agree'].min(axis=0), 3).tolist()}); max abs error "
          f"{({k_: float(f'{v:.2e}') for k_, v in errs.items()})} (reference self-drift {np.round(g['self_drift'], 5).tolist()})")
    report = os.environ.get("HSP_REPORT_DIR")
    if report:
        import json
        with open(os.path.join(report, f"free_running_{name}.json"), "w") as f:
            json.dump({"agree_rows_per_layer": watch.agree, "agree_sets_per_layer": watch.agree_set, "max_abs_err": errs,
                       "reference_self_agree": g["self_agree"].tolist(),
                       "reference_self_drift": g["self_drift"].tolist()}, f, indent=1)
    assert all(a >= f for a, f in zip(watch.agree, REFINIT_AGREE[name])), watch.agree
    for n_, e in errs.items():
        assert e <= REFINIT_BOUND[name], f"{name} {n_}: {e:.3e} > {REFINIT_BOUND[name]}"


def test_stack_tiled_trainbn(dev, ref, flags, monkeypatch):
    """TRAINING-mode rule on tiled clouds (400 / 1000 / 257 / 600 base points padded to 1028 by repetition, datasets/load_data.py:
    314-316): the coordinate searches follow torch.topk's order among equal distances in training too (ops.knn_xyz is the only
    route), so with the reference's FEATURE-space lists replayed (train-mode BatchNorm rules out bit-equal feature rows) the
    forward meets the reference at 1e-4 and the backward of the HS stack lands inside the bounds of the tie-free case -- the max
    over a neighbourhood that holds a point AND its duplicate ties exactly there, and the gradient must go where torch.max's
    first-index rule sends it.  Fixture: oracle/gen_golden_tiled.py (reference-initialised weights)."""
    from conftest import tiled_batch
    from hs_pose_amd import ops
    from hs_pose_amd.PoseNet9D import PoseNet9D
    g = golden("stack_tiled_trainbn_1028")
    B, N, seed = (int(v) for v in g["meta"][:3])
    bases = [int(v) for v in g["meta"][4:]]

    def build():
        flags.train = 0
        torch.manual_seed(0)
        net = PoseNet9D().to(dev)
        net.train(True)
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        return net
    pts = tiled_batch(ref, bases, seed, N).to(dev)
    obj = torch.from_numpy((ref.hash_unit(B, seed + 1) * 6).astype(np.int64)).float().view(B, 1).to(dev)
    xyz_lists = {}
    real_xyz = ops.knn_xyz

    def rec_xyz(x, k, k2=0, drop_first=True):
        a, b = real_xyz(x, k, k2, drop_first)
        xyz_lists[(x.shape[1], k)] = a
        if b is not None:
            xyz_lists[(x.shape[1], k2)] = b
        return a, b
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
    forced = ForcedFeatKnn(monkeypatch, g, dev)
    net = build()
    torch.manual_seed(1)
    with torch.no_grad():
        outs = dict(zip(OUT_NAMES, net(pts, obj)))
    for (n_, k_) in ((1028, 20), (1028, 4), (257, 20), (257, 4), (64, 8)):
        rows = float((xyz_lists[(n_, k_)].cpu().numpy() == g[f"xyz_n{n_}_k{k_}"]).all(-1).mean())
        assert rows == 1.0, f"xyz search N = {n_}, k = {k_}: {rows:.4f} of the rows carry the reference's ordered list"
    errs = {n_: _maxerr(outs[n_], g["out." + n_]) for n_ in OUT_NAMES[4:]}
    print(f"TILED, train-mode BatchNorm, feature lists replayed: own feature lists equal to the reference's on {forced.agree} of the "
          f"rows per HS layer (sets {forced.agree_set}); max abs error {({k_: float(f'{v:.2e}') for k_, v in errs.items()})}")
    for n_, e in errs.items():
        assert e <= 1e-4, f"{n_}: {e:.3e}"
    # unit U1 backward from the fixture's centred cloud
    net = build()
    torch.manual_seed(1)
    _, _, feat = net.face_recon(torch.from_numpy(g["centred"]).to(dev), obj)
    feat = feat[..., :1286]
    scale = max(1.0, np.abs(g["feat"]).max())
    assert _maxerr(feat.reshape(-1)[::211], g["feat"]) <= 1e-4 * scale
    dfeat = ref.hash_tensor((B, N, 1286), seed + 5, 1.0).to(dev)
    (feat * dfeat).sum().backward()
    checked = 0
    worst = {}
    for pn, p in net.face_recon.named_parameters():
        key = "gradsample." + pn
        if key not in g.files:
            continue
        want = g[key]
        norm, _ = g["gradnorm." + pn]
        got = p.grad.reshape(-1)[::499].cpu().double().numpy()
        last = pn.startswith("conv_4.")
        tol = 3e-2 * max(np.abs(want).max(), norm / max(p.numel(), 1) ** 0.5, 1e-12)
        assert np.abs(got - want).max() <= tol, f"{pn}: {np.abs(got - want).max():.3e} > {tol:.3e}"
        gn = p.grad.double().norm().item()
        worst[pn] = abs(gn - norm) / max(norm, 1e-12)
        assert abs(gn - norm) <= (1e-4 if last else 3e-3) * max(norm, 1e-12), f"{pn}: grad norm {gn} vs {norm}"
        checked += 1
    assert checked >= 26
    print("TILED backward: worst relative gradient-norm error", max(worst.items(), key=lambda kv: kv[1]))
    for bn_ in ("bn1", "bn2", "bn3"):
        for st in ("running_mean", "running_var"):
            want = g[f"bnstat.{bn_}.{st}"]
            got = getattr(getattr(net.face_recon, bn_), st).cpu().numpy()
            assert np.abs(got - want).max() <= 1e-4 * max(1.0, np.abs(want).max()), (bn_, st)
