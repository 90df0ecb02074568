"""CPU: the oracle (oracle/hsp_oracle.c + oracle/ref_cpu.py) against the reference's golden vectors
(tests/golden/*.npz, produced by oracle/gen_golden.py from the imported reference)."""
import numpy as np
import pytest
import torch

from conftest import golden


@pytest.fixture(autouse=True)
def _one_thread():
    """the fixtures are written with ONE torch thread (oracle/gen_golden.py: bit-reproducible regeneration); the
    restatement is held to 1e-6 here, which threaded MKL reductions + train-mode BN on 4 rows can exceed"""
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


KNN_CASES = ["knn_xyz_1028", "knn_xyz_257", "knn_xyz_64_k8", "knn_xyz_1028_k4", "knn_feat128_1028",
             "knn_feat128_257", "knn_feat256_257", "knn_feat256_64_k8", "knn_feat16_128_k8", "knn_feat32_16_k2",
             "knn_relu_feat128_257"]


def _case_input(ref, g):
    meta = g["meta"]
    shape, seed, k = tuple(int(v) for v in meta[:3]), int(meta[3]), int(meta[4])
    scale, off = (float(v) for v in g["scale_off"])
    return ref.hash_tensor(shape, seed, scale, off), k


def test_config1_known_answer(ref, oc):
    g = golden("knn_cfg1")
    x = torch.from_numpy(g["x"])
    assert ref.knn_index(x, 20)[0, 0, :5].tolist() == [121, 239, 166, 46, 230]
    assert np.array_equal(ref.knn_index(x, 20).numpy(), g["idx"].astype(np.int64))
    assert np.array_equal(oc.knn(g["x"], 20), g["idx"].astype(np.int32))


@pytest.mark.parametrize("name", KNN_CASES)
def test_knn_oracles_match_reference(ref, oc, name):
    g = golden(name)
    x, k = _case_input(ref, g)
    if "relu" in name:
        x = torch.relu(x)
    want = g["idx"].astype(np.int32)
    assert np.array_equal(oc.knn(x.numpy(), k), want)                # C oracle: bit-exact indices
    assert np.array_equal(ref.knn_index(x, k).numpy(), want)         # torch restatement


@pytest.mark.parametrize("name", ["exact_stack_tiled_1028", "stack_tiled_trainbn_1028"])
def test_knn_oracles_match_reference_on_tiled_clouds(ref, oc, name):
    """tiled clouds (the reference's loader pads a short crop by repetition: exact duplicates, ties everywhere): the five xyz
    neighbour lists the reference's forward computed (oracle/gen_golden_tiled.py) against the C oracle's torch.topk restatement
    and the torch restatement -- including Pool_layer's k = 4 list, which is NOT the prefix of the k = 20 list there"""
    from conftest import tiled_batch
    g = golden(name)
    B, N, seed = (int(v) for v in g["meta"][:3])
    bases = [int(v) for v in g["meta"][4:]]
    pts = tiled_batch(ref, bases, seed, N)
    centred = pts - pts.mean(dim=1, keepdim=True)                  # PoseNet9D.py:25
    if "centred" in g.files:
        assert np.array_equal(centred.numpy(), g["centred"])
    for k in (20, 4):
        want = g[f"xyz_n{N}_k{k}"].astype(np.int32)
        assert np.array_equal(oc.knn_topk(centred.numpy(), k), want), k
        assert np.array_equal(ref.knn_index(centred, k).numpy().astype(np.int32), want), k
    l20, l4 = g[f"xyz_n{N}_k20"], g[f"xyz_n{N}_k4"]
    differ = float((l20[:, :, :4] != l4).any(-1).mean())
    assert differ > 0.3 and abs(differ - float(g["k4_vs_k20_prefix"][0])) < 1e-6
    # the lowest-index rule (hsp_oracle_knn) is NOT the reference's rule on such a cloud
    assert not np.array_equal(oc.knn(centred.numpy(), 4), l4.astype(np.int32))
    # the coarser levels: the pooled vertices are rows of the fixture
    for n1, k1 in ((257, 20), (257, 4), (64, 8)):
        v = torch.from_numpy(g["pool_1.vertices" if n1 == 257 else "pool_2.vertices"])
        assert np.array_equal(oc.knn_topk(v.numpy(), k1), g[f"xyz_n{n1}_k{k1}"].astype(np.int32)), (n1, k1)


def test_knn_tie_case(ref, oc):
    """exact ties: torch.topk's order is unspecified, so only the selected distance VALUES are pinned."""
    g = golden("knn_xyz_offset")
    x, k = _case_input(ref, g)
    assert int(g["meta"][5]) == 0
    d = ref.knn_dist(x)
    ci = oc.knn(x.numpy(), k)
    dv_ref = torch.gather(d, 2, torch.from_numpy(g["idx"].astype(np.int64))).numpy()
    dv_c = torch.gather(d, 2, torch.from_numpy(ci.astype(np.int64))).numpy()
    assert np.array_equal(dv_ref, dv_c)
    # lowest index first inside every equal-distance run of the C oracle's rows
    same = dv_c[:, :, 1:] == dv_c[:, :, :-1]
    assert (ci[:, :, 1:][same] > ci[:, :, :-1][same]).all()


def test_quad_matches_aten_order(ref, oc):
    for C in (3, 6, 16, 33, 128, 256, 700):
        x = ref.hash_tensor((50, C), C, 1.0)
        want = (x ** 2).sum(dim=1).numpy()
        assert np.array_equal(oc.quad(x.numpy()).view(np.uint32), want.view(np.uint32)), C


@pytest.mark.parametrize("name,m", [("nn1_1028_257", 257), ("nn1_1028_64", 64)])
def test_nearest(ref, oc, name, m):
    g = golden(name)
    tgt = ref.hash_tensor((2, 1028, 3), 31, 0.1)
    src = tgt[:, torch.from_numpy(g["perm"].astype(np.int64)), :].contiguous()
    assert np.array_equal(oc.nn1(tgt.numpy(), src.numpy()), g["idx"].astype(np.int32))
    assert np.array_equal(ref.nearest_index(tgt, src).squeeze(-1).numpy(), g["idx"].astype(np.int64))


def _fill(ref, keys_shapes):
    sd = {k: torch.empty(*shape) if shape else torch.zeros((), dtype=torch.long) for k, shape in keys_shapes.items()}
    for k in sd:
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.long)
    ref.fill_state_closed_form(sd)
    return sd


def test_surface_and_hs_layers(ref):
    g = golden("surface_small")
    K, S, N, k, B, seed = (int(v) for v in g["meta"])
    sd = _fill(ref, {"directions": (3, S * K), "STE_layer.weight": (K, 3, 1), "conv2.weight": (K, 2 * K, 1)})
    p = {k_: v.requires_grad_(True) for k_, v in sd.items()}
    xyz = ref.hash_tensor((B, N, 3), seed, 0.1)
    up = ref.hash_tensor((B, N, K), seed + 1, 1.0)
    out = ref.surface_layer(p, "", xyz, k, S)
    assert np.array_equal(out.detach().numpy(), g["out"])
    (out * up).sum().backward()
    for k_ in p:
        assert np.allclose(p[k_].grad.numpy(), g["grad." + k_], rtol=1e-5, atol=1e-5 * np.abs(g["grad." + k_]).max())

    g = golden("hs_small")
    Cin, Cout, S, N, k, B, seed = (int(v) for v in g["meta"])
    sd = _fill(ref, {"weights": (Cin, (S + 1) * Cout), "bias": ((S + 1) * Cout,), "directions": (3, S * Cout),
                     "STE_layer.weight": (Cout, Cin, 1), "conv2.weight": (Cout, 2 * Cout, 1)})
    p = {k_: v.requires_grad_(True) for k_, v in sd.items()}
    xyz = ref.hash_tensor((B, N, 3), seed, 0.1)
    fmap = torch.relu(ref.hash_tensor((B, N, Cin), seed + 2, 1.0)).requires_grad_(True)
    up = ref.hash_tensor((B, N, Cout), seed + 1, 1.0)
    out, idx = ref.hs_layer(p, "", xyz, fmap, k, S, return_idx=True)
    assert np.array_equal(idx.numpy(), g["knn_idx"].astype(np.int64))
    assert np.array_equal(out.detach().numpy(), g["out"])
    (out * up).sum().backward()
    assert np.allclose(fmap.grad.numpy(), g["grad_fmap"], rtol=1e-5, atol=1e-5 * np.abs(g["grad_fmap"]).max())
    for k_ in p:
        assert np.allclose(p[k_].grad.numpy(), g["grad." + k_], rtol=1e-5, atol=1e-5 * np.abs(g["grad." + k_]).max())


def test_pool_and_generator_consumption(ref):
    g = golden("pool_1028")
    torch.manual_seed(1)
    a, b = ref.draw_pool_indices(1028)
    assert np.array_equal(a.numpy(), g["perm_seed1_a"].astype(np.int64))
    assert np.array_equal(b.numpy(), g["perm_seed1_b"].astype(np.int64))
    xyz = ref.hash_tensor((2, 1028, 3), 61, 0.1)
    fmap = ref.hash_tensor((2, 1028, 32), 62, 1.0)
    vp, fp = ref.pool_layer(xyz, fmap, a)
    assert np.array_equal(vp.numpy(), g["v_pool"]) and np.array_equal(fp.numpy(), g["f_pool"])


@pytest.mark.parametrize("name", ["stack_eval_256", "stack_train_256"])
def test_posenet9d_oracle_vs_reference(ref, state_keys, name):
    """whole-model restatement == reference outputs (bit-equal here: same ATen ops, same order)."""
    g = golden(name)
    train_flag, B, N, seed, bn_training = (int(v) for v in g["meta"])
    sd = _fill(ref, state_keys["train" if train_flag else "eval"])
    pts = ref.hash_tensor((B, N, 3), seed, 0.05)
    pts[:, :, 2] += 0.8
    obj = torch.from_numpy((ref.hash_unit(B, seed + 1) * 6).astype(np.int64)).float().view(B, 1)
    pidx = [torch.from_numpy(g["pool_idx0"].astype(np.int64)), torch.from_numpy(g["pool_idx1"].astype(np.int64))]
    with torch.no_grad():
        o = ref.posenet9d(sd, pts, obj, pidx, train_heads=bool(train_flag), bn_training=bool(bn_training))
    assert np.array_equal((pts - pts.mean(dim=1, keepdim=True)).numpy(), g["centred"])
    for n_ in ("p_green_R", "p_red_R", "f_green_R", "f_red_R", "Pred_T", "Pred_s"):
        assert np.allclose(o[n_].numpy(), g["out." + n_], rtol=0, atol=1e-6), n_
    assert np.allclose(o["feat"].reshape(-1)[::1009].numpy(), g["feat_sample"], rtol=0, atol=1e-6)


def test_fps_oracles(ref, oc):
    g = golden("fps_512_64")
    pts = ref.hash_tensor((512, 3), 81, 1.0)
    assert np.array_equal(oc.fps_f64(pts.double().numpy()[None], 64)[0], g["sel"].astype(np.int32))
    assert np.array_equal(oc.fps_f32(pts.numpy()[None], 64)[0], g["sel"].astype(np.int32))


@pytest.mark.parametrize("name", ["chamfer_100_50", "chamfer_257_1028", "chamfer_ties", "chamfer_1_7"])
def test_chamfer_oracle_equals_reference_extension(ref, oc, name):
    """fixtures written by the reference's OWN chamfer_distance.cpp (compiled as is into oracle/_ref/cd_ref.so, CPU entry
    points forward / backward, .cpp:59-87,114-177; oracle/gen_golden_chamfer_fps.py): the C restatement returns the same
    distances, arg-mins (first minimum wins, also on exact ties) and gradients bit for bit."""
    g = golden(name)
    x1, x2, g1, g2 = ref.chamfer_case(name)
    d1, d2, i1, i2 = oc.chamfer_fwd(x1.numpy(), x2.numpy())
    assert np.array_equal(i1, g["idx1"].astype(np.int32)) and np.array_equal(i2, g["idx2"].astype(np.int32))
    assert np.array_equal(d1, g["dist1"]) and np.array_equal(d2, g["dist2"])
    gx1, gx2 = oc.chamfer_bwd(x1.numpy(), x2.numpy(), i1, i2, g1.numpy(), g2.numpy())
    assert np.array_equal(gx1, g["gx1"]) and np.array_equal(gx2, g["gx2"])


@pytest.mark.parametrize("name", ["fps_512_64", "fps_1028_256", "fps_lattice_512_128", "fps_dups_300_40"])
def test_fps_oracle_equals_reference_helper(ref, oc, name):
    """fixtures written by tools/eval_utils.py:107-119 called on float64 and on float32 arrays (numpy computes in the
    array's dtype, sqrt included): both C restatements equal it; on the perturbed lattice a squared-distance fp32 rule
    would pick other points (recorded in the fixture), so the sqrt is part of the contract."""
    g = golden(name)
    pts, ns = ref.fps_case(name)
    assert np.array_equal(oc.fps_f64(pts[None], ns)[0], g["sel_f64"].astype(np.int32))
    assert np.array_equal(oc.fps_f32(pts[None].astype(np.float32), ns)[0], g["sel_f32"].astype(np.int32))
    if name == "fps_lattice_512_128":
        assert not np.array_equal(g["sel_f32_squared_rule"], g["sel_f32"])


def test_chamfer_oracle_vs_bruteforce(ref, oc):
    """the C restatement of chamfer_distance.cpp:59-177 against an independent torch formulation as well"""
    x1 = ref.hash_tensor((2, 100, 3), 91, 0.5).requires_grad_(True)
    x2 = ref.hash_tensor((2, 50, 3), 92, 0.5).requires_grad_(True)
    d1, d2, i1, i2 = oc.chamfer_fwd(x1.detach().numpy(), x2.detach().numpy())
    w1, w2, j1, j2 = ref.chamfer(x1, x2)
    assert np.array_equal(i1, j1.numpy()) and np.array_equal(i2, j2.numpy())
    assert np.allclose(d1, w1.detach().numpy(), atol=1e-7) and np.allclose(d2, w2.detach().numpy(), atol=1e-7)
    u1, u2 = ref.hash_tensor((2, 100), 93, 1.0), ref.hash_tensor((2, 50), 94, 1.0)
    ((w1 * u1).sum() + (w2 * u2).sum()).backward()
    gx1, gx2 = oc.chamfer_bwd(x1.detach().numpy(), x2.detach().numpy(), i1, i2, u1.numpy(), u2.numpy())
    assert np.allclose(gx1, x1.grad.numpy(), atol=1e-5) and np.allclose(gx2, x2.grad.numpy(), atol=1e-5)


def test_topk_tie_order_matches_torch(oc):
    """torch.topk's order among exactly equal distances = libstdc++'s nth_element + sort / partial_sort under a value-only
    comparator: the C restatement (oracle/hsp_oracle.c, the checker of csrc/knn_exact.hip) against the indices the imported
    reference's torch.topk returned on tie-rich rows (oracle/gen_golden_exact.py), and against this host's torch.topk"""
    g = golden("exact_topk_ties")
    for tag in ("n1028_m21", "n257_m21", "n64_m9", "n1028_m5", "n4096_m21"):
        q, m = (int(v) for v in g["q_" + tag])
        d = g["d_" + tag].astype(np.float32) / np.float32(q)
        got = oc.topk_smallest(d, m)
        assert np.array_equal(got, g["i_" + tag].astype(np.int32)), tag
        assert np.array_equal(got, torch.topk(torch.from_numpy(d), m, dim=-1, largest=False)[1].numpy()), tag
