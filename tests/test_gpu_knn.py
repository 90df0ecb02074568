"""GPU parity: neighbour search through the C-ABI vs the reference's golden indices and the C oracle.
Bar: bit-exact indices (SURVEY 8c / BASELINE north_star "KNN indices bit-exact")."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu

KNN_CASES = ["knn_xyz_1028", "knn_xyz_257", "knn_xyz_64_k8", "knn_xyz_1028_k4", "knn_feat128_1028",
             "knn_feat128_257", "knn_feat256_257", "knn_feat256_64_k8", "knn_feat16_128_k8", "knn_feat32_16_k2",
             "knn_relu_feat128_257"]


def _case_input(ref, g):
    meta = g["meta"]
    shape, seed, k = tuple(int(v) for v in meta[:3]), int(meta[3]), int(meta[4])
    scale, off = (float(v) for v in g["scale_off"])
    return ref.hash_tensor(shape, seed, scale, off), k


def test_config1_known_answer(dev, ref):
    """BASELINE.json configs[0]: B=1 N=256 C=3 cloud, KNN-index parity with the reference CPU path."""
    from hs_pose_amd import gcn3d
    g = golden("knn_cfg1")
    x = torch.from_numpy(g["x"]).to(dev)
    idx = gcn3d.get_neighbor_index(x, 20)
    assert idx.dtype == torch.int64 and idx.shape == (1, 256, 20)
    assert idx[0, 0, :5].tolist() == [121, 239, 166, 46, 230]
    assert np.array_equal(idx.cpu().numpy(), g["idx"].astype(np.int64))


@pytest.mark.parametrize("name", KNN_CASES)
def test_knn_golden_bit_exact(dev, ref, name):
    from hs_pose_amd import ops
    g = golden(name)
    x, k = _case_input(ref, g)
    if "relu" in name:
        x = torch.relu(x)
    idx = ops.knn(x.to(dev), k).cpu().numpy()
    assert np.array_equal(idx, g["idx"].astype(np.int32)), f"{name}: {(idx != g['idx']).sum()} mismatching entries"


def test_knn_ties_value_equal(dev, ref, oc):
    """uncentred cloud with exact fp32 distance ties: a coordinate search returns the REFERENCE's own picks (torch.topk's order
    among equal distances, csrc/knn_exact.hip) -- the fixture written by the imported reference, and the C oracle's restatement."""
    from hs_pose_amd import ops
    g = golden("knn_xyz_offset")
    x, k = _case_input(ref, g)
    idx = ops.knn(x.to(dev), k).cpu().numpy()
    assert np.array_equal(idx, g["idx"].astype(np.int32))
    assert np.array_equal(idx, oc.knn_topk(x.numpy(), k))
    d = ref.knn_dist(x)
    dv_ref = torch.gather(d, 2, torch.from_numpy(g["idx"].astype(np.int64))).numpy()
    dv_gpu = torch.gather(d, 2, torch.from_numpy(idx.astype(np.int64))).numpy()
    assert np.array_equal(dv_ref, dv_gpu)


@pytest.mark.parametrize("B,N,C,k,drop", [
    (1, 21, 3, 20, 1),        # N == k+1: every other point is a neighbour
    (3, 33, 3, 8, 1), (2, 100, 3, 1, 1), (2, 100, 3, 1, 0), (2, 300, 3, 32, 1), (1, 64, 3, 16, 0),
    (40, 1028, 3, 20, 1),     # B*N >= 32768 -> 4 lanes per query
    (130, 1028, 3, 4, 1),     # B*N >= 131072 -> 1 lane per query
    (1, 4100, 3, 20, 1),      # more points than one LDS chunk
    (2, 40, 8, 4, 1), (2, 70, 100, 8, 1), (1, 33, 64, 2, 1), (2, 130, 192, 20, 0), (1, 500, 128, 32, 1),
    (1, 31, 6, 3, 1),
    (16, 1028, 128, 20, 1),   # the bench shape: 32 MFMA tiles + the 4 remainder queries through the symmetric path
    (32, 260, 64, 8, 1),      # remainder queries with K1 = 9
    (3, 1032, 128, 20, 1),    # 8 remainder queries as extra workgroups
    (17, 1032, 64, 20, 1),    # 8 remainder queries, symmetric path (>= 512 tiles): both halves of the partial tile
    (64, 260, 7, 5, 1),       # symmetric path with odd C (generic load path), K1 = 9
    (2, 69, 7, 5, 1),         # 5 remainder queries, odd C (generic load path)
    (2, 72, 64, 20, 1),       # only two query tiles: waves 2, 3 idle, no shared bound
    (2, 4096, 128, 20, 1),    # dense cloud (BASELINE configs[3] point count): 128 query tiles per cloud
])
def test_knn_vs_c_oracle(dev, ref, oc, B, N, C, k, drop):
    from hs_pose_amd import ops
    x = ref.hash_tensor((B, N, C), 1000 + N + C + k, 0.3)
    idx = ops.knn(x.to(dev), k, drop_first=bool(drop)).cpu().numpy()
    # coordinates: torch.topk's order among equal distances; features (outside the exact scope): lowest index first
    want = oc.knn_topk(x.numpy(), k, drop) if C == 3 else oc.knn(x.numpy(), k, drop)
    assert np.array_equal(idx, want), f"{(idx != want).sum()} of {idx.size} differ"


def test_knn_duplicate_points(dev, ref, oc):
    """tiled clouds (the dataset pads short clouds by repetition, datasets/load_data.py:314-316): heavy exact ties;
    'drop rank 0' is not 'exclude self'.  Coordinates: the kernel returns what torch.topk returns on the host (ref.knn_index runs
    the torch ops of gcn3d.py:15-24) == the C oracle's restatement of it, for both branches of ATen's topk (nth_element + sort at
    k = 20, partial_sort at k = 4 once N >= 320).  Feature rows: lowest index first by default, torch.topk's order in the exact scope."""
    from hs_pose_amd import ops
    base = ref.hash_tensor((2, 300, 3), 77, 0.1)
    x = torch.cat([base, base[:, :212]], dim=1).contiguous()      # 512 points, 212 duplicated
    for k in (20, 4):
        idx = ops.knn(x.to(dev), k).cpu().numpy()
        assert np.array_equal(idx, ref.knn_index(x, k).numpy().astype(np.int32)), k
        assert np.array_equal(idx, oc.knn_topk(x.numpy(), k)), k
    i20, i4 = ops.knn_xyz(x.to(dev), 20, 4)                       # one search, both lists
    assert np.array_equal(i20.cpu().numpy(), oc.knn_topk(x.numpy(), 20)) and np.array_equal(i4.cpu().numpy(), oc.knn_topk(x.numpy(), 4))
    assert not torch.equal(i20[:, :, :4], i4)                     # (the short list is not the prefix of the long one here)
    # more coincident points than the wave kernel's survivor scratch: its extraction fallback
    heavy = ref.hash_tensor((2, 400, 3), 80, 0.1)
    heavy[:, 50:250] = heavy[:, 10:11]                            # 201 identical points (all mutual distances tie)
    heavy[:, 300:340] = heavy[:, 20:21]
    idx = ops.knn(heavy.to(dev), 20).cpu().numpy()
    assert np.array_equal(idx, ref.knn_index(heavy, 20).numpy().astype(np.int32))
    idx = ops.knn(heavy.to(dev), 4, drop_first=False).cpu().numpy()
    assert np.array_equal(idx, oc.knn_topk(heavy.numpy(), 4, 0))
    # the plain (distance, index) selection underneath still equals the lowest-index oracle
    idx = ops.knn(heavy.to(dev), 20, _plain_xyz=True).cpu().numpy()
    assert np.array_equal(idx, oc.knn(heavy.numpy(), 20))
    # the per-lane-list kernel (B * N >= 131072) and the small-N form take the same flags
    big = ref.hash_tensor((130, 1028, 3), 81, 0.1)
    big[:, 600:] = big[:, :428]
    idx = ops.knn(big.to(dev), 4).cpu().numpy()
    assert np.array_equal(idx[:3], oc.knn_topk(big[:3].numpy(), 4)) and np.array_equal(idx[-2:], oc.knn_topk(big[-2:].numpy(), 4))
    small = ref.hash_tensor((3, 40, 3), 82, 0.1)
    small[:, 25:] = small[:, :15]
    assert np.array_equal(ops.knn(small.to(dev), 8).cpu().numpy(), oc.knn_topk(small.numpy(), 8))
    xf = torch.relu(ref.hash_tensor((1, 200, 32), 78, 1.0))
    xf = torch.cat([xf, xf[:, :56]], dim=1).contiguous()
    idx = ops.knn(xf.to(dev), 8).cpu().numpy()
    assert np.array_equal(idx, oc.knn(xf.numpy(), 8))
    with ops.exact_scope(True):
        idx = ops.knn(xf.to(dev), 8).cpu().numpy()
    assert np.array_equal(idx, oc.knn_topk(xf.numpy(), 8))
    # duplicates across the remainder rows (N = 260: rows 256..259 repeat rows 0..3) and a shared bound full of ties
    xg = torch.relu(ref.hash_tensor((2, 256, 64), 79, 1.0))
    xg = torch.cat([xg, xg[:, :4]], dim=1).contiguous()
    xg[:, 100:140] = xg[:, 60:61]                                  # 41 identical rows: ties at distance 0
    idx = ops.knn(xg.to(dev), 20).cpu().numpy()
    assert np.array_equal(idx, oc.knn(xg.numpy(), 20))


@pytest.mark.parametrize("B,N,C,k,drop", [
    (16, 257, 128, 20, 1), (16, 257, 256, 20, 1), (16, 64, 256, 8, 1),     # the stack's coarse levels (QT = 17 / 17 / 4)
    (1, 257, 128, 20, 1), (1, 64, 256, 8, 1),                              # one image instance: QT = 2
    (3, 65, 64, 20, 1), (5, 68, 192, 8, 0),                                # 1 / 4 remainder rows as query columns, one full tile
    (7, 260, 128, 20, 1), (2, 259, 64, 5, 1),                              # remainder rows owned by several workgroups
    (4, 80, 128, 8, 1), (9, 300, 64, 20, 1), (2, 320, 448, 31, 1),         # long remainders: a padded MFMA tile; C = 448; K = 32 lists
    (40, 96, 128, 20, 1), (300, 64, 64, 8, 1),                             # more clouds than CUs: QT = 32
])
def test_knn_small_cloud_kernel(dev, ref, oc, B, N, C, k, drop):
    """knn_feat_small_kernel (64 <= N <= 320, C % 64 == 0, C < 512): one wave per candidate tile, |x|^2 in the kernel, the remainder
    rows as query columns, register selection two queries at a time -- every branch of its shape logic against the C oracle, plus the
    exact scope's flags + distance matrix on the same shapes (torch.topk's order)."""
    from hs_pose_amd import ops
    x = torch.relu(ref.hash_tensor((B, N, C), 4000 + N + C + k, 0.3))
    idx = ops.knn(x.to(dev), k, drop_first=bool(drop)).cpu().numpy()
    want = oc.knn(x.numpy(), k, drop)
    assert np.array_equal(idx, want), f"{(idx != want).sum()} of {idx.size} differ"
    if B <= 16 and k + drop + 1 <= 33:
        # duplicated rows (a tiled cloud's features): ties in every row, some rows with more survivors than the 64 register slots
        xt = x[:min(B, 3)].clone()
        xt[:, N // 2:] = xt[:, :N - N // 2].clone()
        xt[:, 5:5 + min(70, N // 3)] = xt[:, 3:4].clone()
        idx = ops.knn(xt.to(dev), k, drop_first=bool(drop)).cpu().numpy()
        assert np.array_equal(idx, oc.knn(xt.numpy(), k, drop))
        with ops.exact_scope(True):
            idx = ops.knn(xt.to(dev), k, drop_first=bool(drop)).cpu().numpy()
        assert np.array_equal(idx, oc.knn_topk(xt.numpy(), k, drop))


@pytest.mark.parametrize("name", ["nn1_1028_257", "nn1_1028_64"])
def test_nn1_golden(dev, ref, name):
    from hs_pose_amd import gcn3d
    g = golden(name)
    tgt = ref.hash_tensor((2, 1028, 3), 31, 0.1)
    src = tgt[:, torch.from_numpy(g["perm"].astype(np.int64)), :].contiguous()
    idx = gcn3d.get_nearest_index(tgt.to(dev), src.to(dev))
    assert idx.shape == (2, 1028, 1) and idx.dtype == torch.int64
    assert np.array_equal(idx.squeeze(-1).cpu().numpy(), g["idx"].astype(np.int64))


@pytest.mark.parametrize("name", ["exact_stack_tiled_1028", "stack_tiled_trainbn_1028"])
def test_knn_xyz_tiled_lists(dev, ref, name):
    """the five coordinate searches of a forward on TILED clouds against the lists the imported reference computed
    (oracle/gen_golden_tiled.py): k = 20 and Pool_layer's k = 4 at N = 1028 and 257 from one call each, k = 8 at N = 64"""
    from conftest import tiled_batch
    from hs_pose_amd import ops
    g = golden(name)
    B, N, seed = (int(v) for v in g["meta"][:3])
    pts = tiled_batch(ref, [int(v) for v in g["meta"][4:]], seed, N)
    centred = torch.from_numpy(g["centred"]) if "centred" in g.files else pts - pts.mean(dim=1, keepdim=True)
    i20, i4 = ops.knn_xyz(centred.to(dev), 20, 4)
    assert np.array_equal(i20.cpu().numpy(), g[f"xyz_n{N}_k20"]) and np.array_equal(i4.cpu().numpy(), g[f"xyz_n{N}_k4"])
    v1 = torch.from_numpy(g["pool_1.vertices"]).to(dev)
    i20, i4 = ops.knn_xyz(v1, 20, 4)
    assert np.array_equal(i20.cpu().numpy(), g["xyz_n257_k20"]) and np.array_equal(i4.cpu().numpy(), g["xyz_n257_k4"])
    v2 = torch.from_numpy(g["pool_2.vertices"]).to(dev)
    assert np.array_equal(ops.knn(v2, 8).cpu().numpy(), g["xyz_n64_k8"])


def test_knn_full_size_properties(dev, ref):
    """BASELINE configs[1] size (B=16, N=1028): size-independent properties instead of a CPU oracle run:
    ascending distances, no duplicates in a row, k=4 list == prefix of k=20 list, permutation
    equivariance (renumbering the points renumbers the neighbours)."""
    from hs_pose_amd import ops
    x = ref.hash_tensor((16, 1028, 3), 5, 0.05).to(dev)
    idx = ops.knn(x, 20).long()
    d = ((x.unsqueeze(2) - torch.gather(x.unsqueeze(1).expand(-1, 1028, -1, -1), 2, idx.unsqueeze(-1).expand(-1, -1, -1, 3))) ** 2).sum(-1)
    assert (d[:, :, 1:] - d[:, :, :-1] >= -1e-9).all()
    s = idx.sort(dim=2)[0]
    assert (s[:, :, 1:] != s[:, :, :-1]).all()
    assert torch.equal(ops.knn(x, 4).long(), idx[:, :, :4])
    perm = torch.randperm(1028, generator=torch.Generator().manual_seed(3)).to(dev)
    inv = torch.empty_like(perm); inv[perm] = torch.arange(1028, device=dev)
    idx_p = ops.knn(x[:, perm].contiguous(), 20).long()
    # neighbour sets must agree (order too, barring exact ties whose index order changes with numbering)
    back = perm[idx_p][:, inv]
    same = (back.sort(dim=2)[0] == idx.sort(dim=2)[0]).all(dim=2).float().mean().item()
    assert same > 0.999


@pytest.mark.parametrize("B,N,C,k", [(2, 1028, 3, 20), (2, 257, 3, 20), (1, 48, 3, 8), (2, 1028, 128, 20),
                                     (2, 257, 256, 20), (2, 260, 64, 20), (1, 4096, 3, 20)])
def test_knn_non_finite_rows_give_valid_indices(dev, ref, B, N, C, k):
    """NaN / Inf input rows (diverging training, a bad depth pixel) must never become out-of-range neighbour indices:
    the consumers (rf conv gathers, CSR builds) use them unclamped.  Rows without non-finite values keep exact results
    w.r.t. the finite rows; every index stays in [0, N)."""
    from hs_pose_amd import ops
    x = ref.hash_tensor((B, N, C), 123, 1.0)
    bad = [3, N // 2, N - 1]
    x[0, bad[0]] = float("nan")
    x[0, bad[1], 0] = float("inf")
    x[-1, bad[2], C - 1] = float("-inf")
    idx = ops.knn(x.to(dev), k)
    # downstream consumers of the index run without faulting
    f = torch.zeros(B, N, 8, device=dev)
    ops.gather_max(f, idx, k)
    torch.cuda.synchronize()
    idx = idx.cpu().numpy()
    assert idx.min() >= 0 and idx.max() < N
    # a cloud made of NaN only: still valid indices
    idx = ops.knn(torch.full((1, N, C), float("nan"), device=dev), k).cpu().numpy()
    assert idx.min() >= 0 and idx.max() < N


@pytest.mark.parametrize("B,N0,tiled", [(16, 1028, False), (3, 1028, True), (2, 1100, True), (2, 2304, False), (2, 2304, True)])
def test_geometry_levels_equals_the_separate_searches(dev, ref, B, N0, tiled):
    """hsp_geometry_levels_f32 (the coarse levels' vertices, neighbour lists and up-sampling maps as block ranges of one launch)
    against the calls it replaces: rows gathered by torch, ops.knn_xyz / ops.knn per level, ops.nn1 -- bit for bit, on a random
    cloud and on a tiled one (duplicates: the in-kernel tie replay, both branches of ATen's topk)"""
    from hs_pose_amd import ops
    if tiled:
        x = ref.tiled_batch([N0 * 2 // 5] * B, 700 + N0, N0)
    else:
        x = ref.hash_tensor((B, N0, 3), 701 + N0, 0.05)
    x = (x - x.mean(dim=1, keepdim=True)).to(dev)
    n1, n2 = N0 // 4, N0 // 16
    g = torch.Generator().manual_seed(N0)
    sel1 = torch.randperm(N0, generator=g)[:n1].to(device=dev, dtype=torch.int32)
    sel2 = torch.randperm(n1, generator=g)[:n2].to(device=dev, dtype=torch.int32)
    k1, k2 = min(20, n1 // 8), min(20, n2 // 8)
    geo = ops.geometry_levels(x, sel1, sel2, k1, 4, k2)
    assert geo is not None
    v1 = x[:, sel1.long()].contiguous()
    v2 = v1[:, sel2.long()].contiguous()
    assert torch.equal(geo["v1"], v1) and torch.equal(geo["v2"], v2)
    i1, i1p = ops.knn_xyz(v1, k1, 4)
    assert torch.equal(geo["idx1"], i1) and torch.equal(geo["idx1_pool"], i1p)
    assert torch.equal(geo["idx2"], ops.knn(v2, k2))
    assert torch.equal(geo["up1"], ops.nn1(x, v1)) and torch.equal(geo["up2"], ops.nn1(x, v2))
    if tiled:
        assert not torch.equal(i1[:, :, :4], i1p) or n1 < 320
    # the same with the input cloud's own search in the pair of launches (its tie pass rides in the levels' launch)
    geo2 = ops.geometry_all(x, 20, 4, sel1, sel2, k1, 4, k2)
    if 576 < N0 <= 1088:
        i0, i0p = ops.knn_xyz(x, 20, 4)
        assert torch.equal(geo2["idx0"], i0) and torch.equal(geo2["idx0_pool"], i0p)
        for key in ("v1", "v2", "idx1", "idx1_pool", "idx2", "up1", "up2"):
            assert torch.equal(geo2[key], geo[key]), key
    else:
        assert geo2 is None
    # outside the fused kernel's range the caller is told so
    assert ops.geometry_levels(x[:, :200].contiguous(), sel1[:50].contiguous() % 200, sel2[:12].contiguous() % 50, 6, 4, 1) is None
