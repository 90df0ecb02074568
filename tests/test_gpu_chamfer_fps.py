"""GPU parity: Chamfer distance and farthest point sampling vs the C oracle / golden indices."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,n,m", [(2, 100, 50), (1, 1028, 1028), (3, 7, 2500), (2, 2049, 3)])
def test_chamfer_fwd_bwd(dev, ref, oc, B, n, m):
    from hs_pose_amd import ops
    x1 = ref.hash_tensor((B, n, 3), 91, 0.5)
    x2 = ref.hash_tensor((B, m, 3), 92, 0.5)
    d1, d2, i1, i2 = oc.chamfer_fwd(x1.numpy(), x2.numpy())
    a = x1.to(dev).requires_grad_(True)
    b = x2.to(dev).requires_grad_(True)
    g1, g2, j1, j2 = ops.chamfer(a, b)
    assert np.array_equal(j1.cpu().numpy(), i1) and np.array_equal(j2.cpu().numpy(), i2)
    assert np.array_equal(g1.detach().cpu().numpy(), d1) and np.array_equal(g2.detach().cpu().numpy(), d2)   # same fp32 ops: exact
    u1 = ref.hash_tensor((B, n), 93, 1.0)
    u2 = ref.hash_tensor((B, m), 94, 1.0)
    ((g1 * u1.to(dev)).sum() + (g2 * u2.to(dev)).sum()).backward()
    gx1, gx2 = oc.chamfer_bwd(x1.numpy(), x2.numpy(), i1, i2, u1.numpy(), u2.numpy())
    for got, want in ((a.grad, gx1), (b.grad, gx2)):
        err = np.abs(got.cpu().numpy() - want).max()
        assert err <= 1e-5 * max(1.0, np.abs(want).max()), err     # atomics: order differs, values agree


@pytest.mark.parametrize("name", ["chamfer_100_50", "chamfer_257_1028", "chamfer_ties", "chamfer_1_7"])
def test_chamfer_reference_extension_golden(dev, ref, name):
    """against fixtures written by the reference's own chamfer_distance.cpp (oracle/_ref/cd_ref.so, CPU entry points):
    distances and arg-mins bit-exact (same fp32 expression, strict <: first minimum, also on the exact ties of
    chamfer_ties), gradients to 1e-6 of scale (the kernel's atomics sum in another order than the serial scatter)."""
    from hs_pose_amd import ops
    g = golden(name)
    x1, x2, u1, u2 = ref.chamfer_case(name)
    a, b = x1.to(dev).requires_grad_(True), x2.to(dev).requires_grad_(True)
    d1, d2, i1, i2 = ops.chamfer(a, b)
    assert np.array_equal(i1.cpu().numpy(), g["idx1"].astype(np.int32)) and np.array_equal(i2.cpu().numpy(), g["idx2"].astype(np.int32))
    assert np.array_equal(d1.detach().cpu().numpy(), g["dist1"]) and np.array_equal(d2.detach().cpu().numpy(), g["dist2"])
    ((d1 * u1.to(dev)).sum() + (d2 * u2.to(dev)).sum()).backward()
    for got, want in ((a.grad, g["gx1"]), (b.grad, g["gx2"])):
        err = np.abs(got.cpu().numpy() - want).max()
        assert err <= 1e-6 * max(1.0, np.abs(want).max()), (name, err)


@pytest.mark.parametrize("name", ["fps_512_64", "fps_1028_256", "fps_lattice_512_128", "fps_dups_300_40"])
def test_fps_reference_helper_golden(dev, ref, name):
    """the picks of tools/eval_utils.py:107-119 on the float32 cloud (hsp_fps_f32) and on the float64 cloud
    (hsp_fps_f64), bit-exact -- including the perturbed lattice, where only the sqrt'ed-distance rule gets them"""
    from hs_pose_amd import ops
    g = golden(name)
    pts, ns = ref.fps_case(name)
    p64 = torch.from_numpy(pts).unsqueeze(0)
    assert np.array_equal(ops.fps(p64.float().to(dev), ns).cpu().numpy()[0], g["sel_f32"].astype(np.int32))
    assert np.array_equal(ops.fps(p64.to(dev), ns).cpu().numpy()[0], g["sel_f64"].astype(np.int32))
    two = torch.cat([p64, p64.flip(1)], dim=0).contiguous()                   # batched: clouds are independent
    sel = ops.fps(two.to(dev), ns).cpu().numpy()
    assert np.array_equal(sel[0], g["sel_f64"].astype(np.int32)) and sel[1][0] == 0


def test_chamfer_module_surface(dev, ref):
    from hs_pose_amd.chamfer import ChamferDistance
    a = ref.hash_tensor((1, 100, 3), 95, 1.0).to(dev)
    b = ref.hash_tensor((1, 50, 3), 96, 1.0).to(dev)
    d1, d2 = ChamferDistance()(a, b)
    w1, w2, _, _ = ref.chamfer(a.cpu(), b.cpu())
    assert torch.allclose(d1.cpu(), w1, atol=1e-6) and torch.allclose(d2.cpu(), w2, atol=1e-6)
    assert d1.shape == (1, 100) and d2.shape == (1, 50)


def test_fps_golden_and_oracle(dev, ref, oc):
    from hs_pose_amd import ops
    g = golden("fps_512_64")
    pts = ref.hash_tensor((512, 3), 81, 1.0).unsqueeze(0)
    sel = ops.fps(pts.to(dev), 64).cpu().numpy()
    assert np.array_equal(sel[0], g["sel_f32"].astype(np.int32))      # == the reference's numpy helper on this fp32 cloud
    pts = ref.hash_tensor((5, 1028, 3), 82, 0.2)
    assert np.array_equal(ops.fps(pts.to(dev), 257).cpu().numpy(), oc.fps_f32(pts.numpy(), 257))
    pts = ref.hash_tensor((2, 3000, 3), 83, 0.2)
    assert np.array_equal(ops.fps(pts.to(dev), 100).cpu().numpy(), oc.fps_f32(pts.numpy(), 100))


@pytest.mark.parametrize("B,N,m", [(3, 50, 50), (2, 128, 40), (2, 129, 129), (16, 1028, 256), (2, 1281, 300), (2, 2048, 128),
                                   (3, 4096, 1024), (1, 5000, 200), (1, 12288, 64), (1, 13000, 64)])
def test_fps_every_launch_shape(dev, ref, oc, B, N, m):
    """each register-resident instantiation (threads x points per thread), the LDS limit and the generic kernel behind it:
    bit-exact picks against the C oracle"""
    from hs_pose_amd import ops
    pts = ref.hash_tensor((B, N, 3), 84 + N, 0.2)
    assert np.array_equal(ops.fps(pts.to(dev), m).cpu().numpy(), oc.fps_f32(pts.numpy(), m))


def test_fps_duplicate_points(dev, ref, oc):
    """tiled clouds: after the distinct points are exhausted every distance-to-set is 0 and the first index wins"""
    from hs_pose_amd import ops
    base = ref.hash_tensor((2, 100, 3), 90, 0.2)
    pts = torch.cat([base, base, base[:, :56]], dim=1).contiguous()         # 256 points, 100 distinct
    assert np.array_equal(ops.fps(pts.to(dev), 180).cpu().numpy(), oc.fps_f32(pts.numpy(), 180))
