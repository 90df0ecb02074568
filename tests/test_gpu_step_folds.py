"""hsp_step_fold: every fold a backward pass leaves pending (split-K parameter gradients, per-cloud direction-gradient partials)
in ONE launch -- bit-equal to the stand-alone folds (same summation order), through the C-ABI and through ops.StepFolds."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rf_case(ref, dev, B, N, k, S, C, seed):
    from hs_pose_amd import ops
    from hs_pose_amd.ops import _p, _run, _stream
    SC = S * C
    xyz = ref.hash_tensor((B, N, 3), seed, 0.5).to(dev)
    fm = ref.hash_tensor((B, N, (S + 1) * C), seed + 1, 1.0).to(dev)
    dirs = ref.hash_tensor((3, SC), seed + 2, 1.0).to(dev)
    g = ref.hash_tensor((B, N, C), seed + 3, 1.0).to(dev)
    idx = ops.knn(xyz, k)
    out = torch.empty(B, N, C, device=dev)
    arg = torch.empty(B, N, SC, dtype=torch.uint16, device=dev)
    fwin = torch.empty(B, N, SC, device=dev)
    _run("hsp_rf_conv_fwd", (_p(xyz), _p(idx), _p(dirs), _p(fm), B, N, k, S, C, _p(out), _p(arg), _p(fwin), _stream()))
    return xyz, dirs, fwin, arg, g


def test_step_fold_equals_standalone_folds(dev, ref):
    """three receptive-field backwards (two HS layers of different width, one surface layer) and four weight gradients
    (one with a column sum, one into a column block of a wider matrix): partial launches + ONE hsp_step_fold == the complete
    entry points, bit for bit"""
    from hs_pose_amd.ops import _p, _run, _stream, _ws
    from hs_pose_amd._lib import lib, HspWgradPending, HspDirsPending
    L = lib()
    want_d, pend_d, keep = [], [], []
    for (B, N, k, S, C, seed) in ((3, 300, 8, 7, 64, 700), (17, 129, 6, 7, 32, 710)):          # (17 clouds: > 16 slices)
        xyz, dirs, fwin, arg, g = _rf_case(ref, dev, B, N, k, S, C, seed)
        SC = S * C
        wsb = L.hsp_rf_bwd_scatter_workspace_bytes(B, SC)
        gfm0, gd0 = torch.empty(B, N, (S + 1) * C, device=dev), torch.empty(3, SC, device=dev)
        _run("hsp_rf_conv_bwd_scatter", (_p(xyz), _p(dirs), _p(None), _p(fwin), _p(arg), _p(g), B, N, S, C, _p(gfm0), _p(gd0),
                                         _p(_ws(wsb, dev)), wsb, _stream()))
        gfm1, gd1 = torch.empty_like(gfm0), torch.full_like(gd0, float("nan"))
        ws = _ws(wsb, dev)
        p = HspDirsPending()
        _run("hsp_rf_conv_bwd_scatter_partial", (_p(xyz), _p(dirs), _p(None), _p(fwin), _p(arg), _p(g), B, N, S, C, _p(gfm1),
                                                 _p(gd1), _p(ws), wsb, ctypes.byref(p), _stream()))
        assert p.nparts == B and p.SC == SC
        assert torch.equal(gfm0, gfm1)
        want_d.append((gd0, gd1)); pend_d.append(p); keep += [ws, dirs, xyz]
        # the surface form on the same cloud
        gs0, gs1 = torch.empty(3, SC, device=dev), torch.full((3, SC), float("nan"), device=dev)
        _run("hsp_rf_surface_bwd", (_p(xyz), _p(dirs), _p(arg), _p(g), B, N, S, C, _p(gs0), _p(_ws(wsb, dev)), wsb, _stream()))
        ws2 = _ws(wsb, dev)
        p2 = HspDirsPending()
        _run("hsp_rf_surface_bwd_partial", (_p(xyz), _p(dirs), _p(arg), _p(g), B, N, S, C, _p(gs1), _p(ws2), wsb, ctypes.byref(p2),
                                            _stream()))
        want_d.append((gs0, gs1)); pend_d.append(p2); keep.append(ws2)
    want_w, pend_w = [], []
    for i, (K, M, N, colsum, ldc_pad) in enumerate(((5000, 128, 1024, True, 0), (16448, 64, 128, False, 64), (777, 256, 64, True, 0),
                                                    (40000, 128, 128, False, 0))):
        A = ref.hash_tensor((K, M), 800 + i, 1.0).to(dev)
        Bm = ref.hash_tensor((K, N), 810 + i, 1.0).to(dev)
        full0 = torch.zeros(M, N + ldc_pad, device=dev)
        full1 = torch.zeros(M, N + ldc_pad, device=dev)
        c0, c1 = full0[:, ldc_pad:], full1[:, ldc_pad:]
        cs0 = torch.empty(N, device=dev) if colsum else None
        cs1 = torch.full((N,), float("nan"), device=dev) if colsum else None
        wsb = L.hsp_wgrad_workspace_bytes(M, N, K)
        _run("hsp_wgrad_f32", (_p(A), M, _p(Bm), N, M, N, K, _p(c0), N + ldc_pad, _p(cs0), _p(_ws(wsb, dev)), wsb, _stream()))
        ws = _ws(wsb, dev)
        p = HspWgradPending()
        _run("hsp_wgrad_partial_f32", (_p(A), M, _p(Bm), N, M, N, K, _p(c1), N + ldc_pad, _p(cs1), _p(ws), wsb, ctypes.byref(p),
                                       _stream()))
        want_w.append((full0, full1, cs0, cs1)); pend_w.append(p); keep += [ws, A, Bm]
    wa = (HspWgradPending * len(pend_w))(*pend_w)
    da = (HspDirsPending * len(pend_d))(*pend_d)
    _run("hsp_step_fold", (wa, len(pend_w), da, len(pend_d), _stream()))
    torch.cuda.synchronize()
    for a, b in want_d:
        assert torch.equal(a, b)
    for f0, f1, cs0, cs1 in want_w:
        assert torch.equal(f0, f1)
        assert cs0 is None or torch.equal(cs0, cs1)
    # only one kind present / nothing pending
    _run("hsp_step_fold", (wa, 2, None, 0, _stream()))
    _run("hsp_step_fold", (None, 0, da, 1, _stream()))
    _run("hsp_step_fold", (None, 0, None, 0, _stream()))
    assert L.hsp_step_fold(wa, 25, da, 0, None) == -1 and L.hsp_step_fold(wa, 1, da, 9, None) == -1
    assert L.hsp_step_fold(None, 1, None, 0, None) == -1


def test_step_folds_scope_on_the_stack(dev):
    """ops.StepFolds around an eager backward of the HS stack (every .grad None first): one hsp_step_fold launch at the exit and
    the same parameter gradients as the plain backward of a twin network"""
    from hs_pose_amd import gcn3d, ops
    from hs_pose_amd.config import FLAGS
    from hs_pose_amd.FaceRecon import FaceRecon
    from hs_pose_amd.graph import draw_pool_indices
    FLAGS.reset()
    FLAGS.train = 0
    B, N = 2, 256
    g = torch.Generator().manual_seed(5)
    pc = torch.randn(B, N, 3, generator=g) * 0.05
    pc = (pc - pc.mean(dim=1, keepdim=True)).to(dev)
    obj = torch.randint(0, 6, (B, 1), generator=g).float().to(dev)
    dfeat = torch.randn(B, N, 1286, generator=g).to(dev)
    nets = []
    for _ in range(2):
        torch.manual_seed(0)
        nets.append(FaceRecon().to(dev).train())
    pool = [p.to(dev).int() for p in draw_pool_indices(N)]
    grads = []
    for net, scoped in zip(nets, (False, True)):
        timer = ops.KernelTimer(only={"hsp_step_fold", "hsp_wgrad_fold"})
        with gcn3d.pool_index_feed([p.clone() for p in pool]):
            _, _, feat = net(pc, obj)
        prev = ops.set_timer(timer)
        try:
            if scoped:
                with ops.StepFolds():
                    feat.backward(dfeat)
            else:
                feat.backward(dfeat)
        finally:
            ops.set_timer(prev)
        torch.cuda.synchronize()
        names = [r[0] for r in timer.records]
        if scoped:
            assert names == ["hsp_step_fold"], names
        else:
            assert "hsp_step_fold" not in names and len(names) >= 4
        grads.append({k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
    assert grads[0].keys() == grads[1].keys() and len(grads[0]) > 20
    gmax = max(v.abs().max().item() for v in grads[0].values())
    for k in grads[0]:
        err = (grads[0][k] - grads[1][k]).abs().max().item()
        assert err <= 2e-5 * gmax, f"{k}: {err:.3e} (max |grad| {gmax:.3e})"
    with pytest.raises(Exception):
        with ops.StepFolds():
            with ops.StepFolds():
                pass
    assert ops.StepFolds.current is None
