"""The thin C++ PyTorch binding (hs_pose_amd/csrc/hsp_torch.cpp -> _hsp_torch.so): the reference extension's surface
(chamfer_distance.cpp:180-185) and the eval-mode layers as one call each, against the ctypes path over the same libhsp.so
symbols -- BIT FOR BIT (the binding adds checks, allocation and the current stream, no arithmetic) -- and against the fixtures
the reference's own extension wrote."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


def _eq(a, b, what):
    assert a.shape == b.shape and a.dtype == b.dtype, (what, a.shape, b.shape, a.dtype, b.dtype)
    assert torch.equal(a, b), f"{what}: binding and ctypes path differ"


@pytest.fixture
def m(dev):
    from hs_pose_amd._ext import ext
    return ext()


@pytest.mark.parametrize("name", ["chamfer_100_50", "chamfer_257_1028", "chamfer_ties", "chamfer_1_7"])
def test_reference_extension_surface(dev, ref, m, name):
    """forward_cuda / backward_cuda with the reference's argument roles (caller-allocated outputs filled in place), against the
    fixtures written by the reference's chamfer_distance.cpp (oracle/_ref/cd_ref.so): distances and arg-mins bit-exact -- ties
    included --, gradients to 1e-6 of scale (atomics sum in another order than the serial scatter)"""
    g = golden(name)
    x1, x2, u1, u2 = ref.chamfer_case(name)
    a, b = x1.to(dev), x2.to(dev)
    B, n, mm = a.shape[0], a.shape[1], b.shape[1]
    d1, d2 = torch.zeros(B, n, device=dev), torch.zeros(B, mm, device=dev)
    i1, i2 = torch.zeros(B, n, dtype=torch.int, device=dev), torch.zeros(B, mm, dtype=torch.int, device=dev)
    m.forward_cuda(a, b, d1, d2, i1, i2)
    assert np.array_equal(i1.cpu().numpy(), g["idx1"].astype(np.int32)) and np.array_equal(i2.cpu().numpy(), g["idx2"].astype(np.int32))
    assert np.array_equal(d1.cpu().numpy(), g["dist1"]) and np.array_equal(d2.cpu().numpy(), g["dist2"])
    gx1, gx2 = torch.zeros_like(a), torch.zeros_like(b)
    m.backward_cuda(a, b, gx1, gx2, u1.to(dev), u2.to(dev), i1, i2)
    for got, want in ((gx1, g["gx1"]), (gx2, g["gx2"])):
        err = np.abs(got.cpu().numpy() - want).max()
        assert err <= 1e-6 * max(1.0, np.abs(want).max()), (name, err)
    # the plain names are the same entry points (there is no CPU implementation behind them)
    e1, e2 = torch.zeros_like(d1), torch.zeros_like(d2)
    j1, j2 = torch.zeros_like(i1), torch.zeros_like(i2)
    m.forward(a, b, e1, e2, j1, j2)
    _eq(e1, d1, "forward"), _eq(j2, i2, "forward idx")


def test_chamfer_module_is_the_binding(dev, ref):
    """hs_pose_amd.chamfer.ChamferDistance == ops.chamfer (ctypes) bit for bit, forward and backward"""
    from hs_pose_amd import ops
    from hs_pose_amd.chamfer import ChamferDistance
    a = ref.hash_tensor((3, 130, 3), 9101, 0.3).to(dev).requires_grad_(True)
    b = ref.hash_tensor((3, 77, 3), 9102, 0.3).to(dev).requires_grad_(True)
    d1, d2 = ChamferDistance()(a, b)
    (d1.sum() * 2 + (d2 * d2).sum()).backward()
    ga, gb = a.grad.clone(), b.grad.clone()
    a.grad = b.grad = None
    e1, e2, _, _ = ops.chamfer(a, b)
    (e1.sum() * 2 + (e2 * e2).sum()).backward()
    _eq(d1, e1, "dist1"), _eq(d2, e2, "dist2")
    np.testing.assert_allclose(ga.cpu().numpy(), a.grad.cpu().numpy(), atol=1e-6)     # (atomics: order of the float adds)
    np.testing.assert_allclose(gb.cpu().numpy(), b.grad.cpu().numpy(), atol=1e-6)


def test_checks_raise(dev, m):
    """what the reference's binding forgot: device, dtype, contiguity and shape checks raise instead of reading stray memory"""
    x = torch.zeros(2, 64, 3, device=dev)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        m.get_neighbor_index(torch.zeros(2, 64, 3), 4)
    with pytest.raises(RuntimeError, match="dtype"):
        m.get_neighbor_index(x.double(), 4)
    with pytest.raises(RuntimeError, match="contiguous"):
        m.get_neighbor_index(torch.zeros(2, 3, 64, device=dev).transpose(1, 2), 4)
    with pytest.raises(RuntimeError, match="k out of range"):
        m.get_neighbor_index(x, 64)
    d = torch.zeros(2, 64, device=dev)
    i = torch.zeros(2, 64, dtype=torch.int, device=dev)
    with pytest.raises(RuntimeError, match="dtype"):
        m.forward_cuda(x, x, d, d, i.long(), i)
    with pytest.raises(RuntimeError, match=r"\(B,m\)"):
        m.forward_cuda(x, x, d, d[:, :32].contiguous(), i, i[:, :32].contiguous())


def test_index_functions(dev, ref, m):
    """get_neighbor_index / get_nearest_index: int64 like the reference's, equal to the ctypes path"""
    from hs_pose_amd import ops
    x = ref.hash_tensor((2, 300, 3), 9201, 0.2).to(dev)
    f = torch.relu(ref.hash_tensor((2, 300, 64), 9202, 1.0)).to(dev)
    for t, k in ((x, 20), (f, 8)):
        got = m.get_neighbor_index(t, k)
        assert got.dtype == torch.int64
        _eq(got, ops.knn(t, k).long(), "get_neighbor_index")
    src = ref.hash_tensor((2, 75, 3), 9203, 0.2).to(dev)
    _eq(m.get_nearest_index(x, src), ops.nn1(x, src).long().unsqueeze(-1), "get_nearest_index")
    with ops.exact_scope(True):
        _eq(m.knn_exact(f, 8, True, False), ops.knn(f, 8), "knn_exact")
        _eq(m.knn_exact(f, 8, True, True), ops.knn(f, 8, transposed_view=True), "knn_exact, transposed view")


def _layer_inputs(ref, dev, B, N, Cin, C, S, seed):
    xyz = ref.hash_tensor((B, N, 3), seed, 0.05).to(dev)
    X = torch.relu(ref.hash_tensor((B, N, Cin), seed + 1, 1.0)).to(dev)
    W = ref.hash_tensor((Cin, (S + 1) * C), seed + 2, 0.05).to(dev)
    b = ref.hash_tensor(((S + 1) * C,), seed + 3, 0.05).to(dev)
    D = ref.hash_tensor((3, S * C), seed + 4, 0.3).to(dev)
    ws = ref.hash_tensor((C, Cin, 1), seed + 5, 0.05).to(dev)
    w2 = ref.hash_tensor((C, 2 * C, 1), seed + 6, 0.05).to(dev)
    return xyz, X, W, b, D, ws, w2


@pytest.mark.parametrize("B,N,Cin,C,k", [(2, 257, 128, 128, 20), (2, 64, 128, 256, 8), (1, 100, 256, 256, 12)])
def test_hs_layer_one_call(dev, ref, m, B, N, Cin, C, k):
    """hs_layer_forward == the autograd node's exact-scope forward (the same launches, issued from C++)"""
    from hs_pose_amd import ops
    S = 7
    xyz, X, W, b, D, ws, w2 = _layer_inputs(ref, dev, B, N, Cin, C, S, 9300 + C)
    with ops.exact_scope(True):
        idx_f, idx_x = ops.knn(X, k), ops.knn(xyz, k)
        want = ops._HSLayer.apply(xyz, X, idx_f, idx_x, k, S, W, b, D, ws, w2)
        got = m.hs_layer_forward(xyz, X, idx_f, idx_x, k, S, W, b, D, ws, w2)
        _eq(got, want, "hs_layer_forward")
        with torch.no_grad():                          # and the route ops.hs_layer takes by itself in an inference forward
            _eq(ops.hs_layer(xyz, X, idx_f, idx_x, k, S, W, b, D, ws, w2), want, "ops.hs_layer under no_grad")


@pytest.mark.parametrize("relu", [False, True])
def test_surface_pool_bn_centre(dev, ref, m, relu):
    from hs_pose_amd import ops
    S, k, C = 7, 20, 128
    xyz, _, _, _, D, _, w2 = _layer_inputs(ref, dev, 2, 300, 128, C, S, 9400)
    ws = ref.hash_tensor((C, 3, 1), 9407, 0.3).to(dev)
    with ops.exact_scope(True):
        idx_x = ops.knn(xyz, k)
        want = ops._SurfaceLayer.apply(xyz, idx_x, k, S, D, ws, w2, relu)
        want = want[0] if relu else want
        _eq(m.surface_layer_forward(xyz, idx_x, k, S, D, ws, w2, relu), want, "surface_layer_forward")
        feat = want
        sel = torch.randperm(300)[:75].to(device=dev, dtype=torch.int32)
        wo, wv = ops._PoolLayer.apply(feat, xyz, idx_x, sel, 4)
        gv, go = m.pool_forward(xyz, feat, idx_x, sel, 4)
        _eq(go, wo, "pool features"), _eq(gv, wv, "pool vertices")
        bn = torch.nn.BatchNorm1d(C).to(dev).eval()
        with torch.no_grad():
            bn.running_mean.copy_(ref.hash_tensor((C,), 9410, 1.0)); bn.running_var.copy_(ref.hash_tensor((C,), 9411, 1.0).abs() + 0.3)
        inv = ops._eval_invstd(bn)
        wy = ops._BNEval.apply(feat, bn.weight, bn.bias, bn.running_mean, bn.running_var, inv, bn.eps, relu)
        _eq(m.bn_eval(feat, bn.running_mean, bn.running_var, inv, bn.weight, bn.bias, bn.eps, relu), wy, "bn_eval")
        with torch.no_grad():
            _eq(ops.bn_relu(feat, bn, relu=relu), wy, "ops.bn_relu under no_grad")
    pts = xyz + torch.tensor([0.0, 0.0, 0.8], device=dev)
    (wl, wm), (gl, gm) = ops.center_cloud(pts), m.center_cloud(pts)
    _eq(gl, wl, "centred cloud"), _eq(gm, wm, "cloud mean")


def test_launches_go_to_the_current_stream(dev, ref, m):
    """the reference's binding launched on stream 0 whatever the caller's stream; this one follows torch's current stream:
    a long kernel queued on a side stream ahead of the call must order before it (same stream), with no device-wide sync"""
    from hs_pose_amd import ops
    x = ref.hash_tensor((4, 1028, 3), 9501, 0.2).to(dev)
    want = ops.knn(x, 20).long()
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        y = x.clone()
        for _ in range(20):
            y = y * 1.0                                 # work queued on the side stream that the call's input depends on
        got = m.get_neighbor_index(y, 20)
    side.synchronize()
    _eq(got, want, "get_neighbor_index on a side stream")


def test_inference_forward_takes_the_binding(dev, ref, monkeypatch):
    """an eval-mode, no-grad FaceRecon forward issues its HS layers through the binding, and equals the autograd-node route"""
    from hs_pose_amd import ops, _ext
    from hs_pose_amd.config import FLAGS
    from hs_pose_amd.FaceRecon import FaceRecon
    FLAGS.reset(); FLAGS.train = 0
    torch.manual_seed(3)
    net = FaceRecon().to(dev).eval()
    pts = ref.hash_tensor((2, 512, 3), 9601, 0.05).to(dev)
    obj = torch.tensor([[1.0], [4.0]], device=dev)
    calls = []
    real = _ext.ext()

    class Spy:
        def __getattr__(self, name):
            calls.append(name)
            return getattr(real, name)
    monkeypatch.setattr(_ext, "_mod", Spy())
    torch.manual_seed(11)
    with torch.no_grad():
        got = net(pts, obj)[2]
    monkeypatch.setattr(_ext, "_mod", real)
    assert calls.count("hs_layer_forward") == 3 and calls.count("surface_layer_forward") == 1, calls   # conv_4 (2C = 1024): python route
    assert calls.count("pool_forward") == 2 and calls.count("bn_eval") == 3, calls
    torch.manual_seed(11)
    want = net(pts, obj)[2]                              # grad enabled: autograd nodes over ctypes
    _eq(got, want.detach(), "feat")


def test_every_tensor_argument_is_checked(dev, ref, m):
    """a host tensor anywhere in an argument list (a CPU running_mean, a CPU STE weight) is an error, not a host pointer handed to a
    kernel; a bias of the wrong length and a too-narrow pool index are refused; and with the binding declared unavailable the
    no-grad inference forms issue the same launches through ctypes (ops._ext_ok)"""
    from hs_pose_amd import gcn3d, ops
    C, S, k = 32, 3, 4
    xyz = ref.hash_tensor((1, 96, 3), 9601, 0.1).to(dev)
    X = torch.relu(ref.hash_tensor((1, 96, C), 9602, 1.0)).to(dev)
    layer = gcn3d.HS_layer(C, C, S).to(dev).eval()
    idx_x, idx_f = ops.knn(xyz, k), ops.knn(X, k)
    args = [xyz, X, idx_f, idx_x, k, S, layer.weights.detach(), layer.bias.detach(), layer.directions.detach(),
            layer.STE_layer.weight.detach(), layer.conv2.weight.detach()]
    for pos in (6, 7, 8, 9, 10):
        bad = list(args)
        bad[pos] = bad[pos].cpu()
        with pytest.raises(RuntimeError, match="expected a tensor on"):
            m.hs_layer_forward(*bad)
    bad = list(args)
    bad[7] = bad[7][:-1].contiguous()
    with pytest.raises(RuntimeError, match="bias"):
        m.hs_layer_forward(*bad)
    bn = torch.nn.BatchNorm1d(C).to(dev).eval()
    with pytest.raises(RuntimeError, match="expected a tensor on"):
        m.bn_eval(X, bn.running_mean.cpu(), bn.running_var, None, bn.weight, bn.bias, bn.eps, False)
    with pytest.raises(RuntimeError, match="invstd"):
        m.bn_eval(X, bn.running_mean, bn.running_var, torch.ones(C - 1, device=dev), bn.weight, bn.bias, bn.eps, False)
    sel = torch.arange(24, dtype=torch.int32, device=dev)
    with pytest.raises(RuntimeError, match="idx must be"):
        m.pool_forward(xyz, X, idx_x[:, :, :2].contiguous(), sel, 4)
    # the ctypes route of the same no-grad forms
    with torch.no_grad(), ops.exact_scope(True):
        a = ops.hs_layer(*args)
        pa = ops.pool_layer(X, xyz, idx_x, sel, 4)
        prev, ops._ext_state = ops._ext_state, False
        try:
            b = ops.hs_layer(*args)
            pb = ops.pool_layer(X, xyz, idx_x, sel, 4)
            cb = ops.center_cloud(xyz)
        finally:
            ops._ext_state = prev
        ca = ops.center_cloud(xyz)
    _eq(a, b, "hs_layer: binding vs ctypes"), _eq(pa[0], pb[0], "pool: binding vs ctypes"), _eq(ca[0], cb[0], "centre: binding vs ctypes")


@pytest.mark.gpu
def test_library_loaded_before_torch_still_launches():
    """the C-ABI library opened BEFORE anything imported torch (what build() followed by smoke() in one process does): its HIP
    runtime must be the one torch initialises -- with two runtimes in the process every launch fails with hipErrorNoDevice"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from hs_pose_amd._lib import lib\n"
            "assert lib().hsp_version() >= 100\n"
            "import torch\n"
            "from hs_pose_amd import ops\n"
            "idx = ops.knn(torch.randn(1, 64, 3, device='cuda:0'), 8)\n"
            "torch.cuda.synchronize(); print('ok', tuple(idx.shape))\n") % root
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok (1, 64, 8)" in out.stdout, out.stderr[-800:]
