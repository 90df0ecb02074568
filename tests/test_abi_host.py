"""CPU: the C-ABI library loads and exports every symbol include/hsp.h declares; argument validation
returns error codes (no compute without a GPU); host-side mirror of the reference surface."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "hsp.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hsp_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from hs_pose_amd import _lib
    syms = _header_symbols()
    assert len(syms) >= 43
    L = _lib.lib()
    for s in syms:
        assert hasattr(L, s), f"libhsp.so lacks {s}"
    assert sorted(_lib.SIGNATURES) == syms, "ctypes table and include/hsp.h disagree"
    assert L.hsp_version() >= 100
    assert L.hsp_error_string(-3) == b"workspace missing or too small"


def test_argument_validation_without_gpu():
    from hs_pose_amd._lib import lib
    L = lib()
    null, one = ctypes.c_void_p(0), ctypes.c_void_p(16)
    assert L.hsp_knn_f32(null, 1, 8, 3, 2, 1, null, null, 0, null) == -1          # null pointers
    assert L.hsp_knn_f32(one, 1, 8, 3, 8, 1, one, null, 0, null) == -1            # k + 1 > N
    assert L.hsp_knn_f32(one, 1, 100, 3, 40, 1, one, null, 0, null) == -1         # k > HSP_MAX_K
    assert L.hsp_knn_f32(one, 1, 100, 64, 4, 1, one, null, 0, null) == -3         # feature path needs workspace
    assert L.hsp_knn_workspace_bytes(2, 100, 3, 4) == 0
    assert L.hsp_knn_workspace_bytes(2, 100, 128, 4) == 2 * 100 * 4
    assert L.hsp_rf_conv_fwd(one, one, one, one, 1, 70000, 4, 7, 128, one, one, null, null) == -2  # N > 65535 (uint16 rows)
    assert L.hsp_rf_conv_fwd(one, one, one, one, 1, 8, 4, 7, 126, one, one, null, null) == -2     # C % 4
    assert L.hsp_rf_conv_bwd(one, one, one, one, one, one, one, 1, 8, 4, 7, 128, one, one, null, 0, null) == -3
    assert L.hsp_rev_build(one, 1, 8, 8, 4, 2, one, one, null) == -1                         # kstride < k
    assert L.hsp_gather_max_bwd_csr(one, 0, one, one, one, 1, 8, 8, 4, 126, one, null) == -2  # C % 4
    assert L.hsp_gather_max_fwd(one, one, null, 1, 8, 8, 4, 2, 2, 16, one, one, null) == -1  # Nq != Nidx w/o qsel
    assert L.hsp_gather_rows_fwd(one, one, 0, 1, 8, 8, 16, one, 8, null) == -1               # out_stride < C
    assert L.hsp_fps_f32(one, 1, 8, 9, one, one, 1024, null) == -1
    assert L.hsp_fps_workspace_bytes(2, 100) == 800
    assert L.hsp_wgrad_f32(one, 128, one, 1024, 100, 1024, 4000, one, 1024, null, one, 1 << 30, null) == -2   # M % 64
    assert L.hsp_wgrad_f32(one, 128, one, 1024, 128, 1024, 4000, one, 1024, null, null, 0, null) == -3
    assert L.hsp_wgrad_workspace_bytes(128, 1024, 16448) > 0
    assert L.hsp_bn_relu_fwd(one, 100, 12, one, one, 1e-5, 0.1, 1, one, one, one, null, null, null, one, 1 << 20, null) == -2   # 256 % (C/4)
    assert L.hsp_bn_relu_fwd(one, 100, 128, one, one, 1e-5, 0.1, 1, one, one, one, null, null, null, null, 0, null) == -3
    assert L.hsp_bn_workspace_bytes(16448, 128) == 499 * 2 * 128 * 4         # 499 row chunks of 33 rows
    assert L.hsp_rf_bwd_workspace_bytes(896) > 0
    assert L.hsp_rf_bwd_scatter_workspace_bytes(16, 896) == 2 * 16 * 3 * 896 * 4   # (two half-cloud tiles on dense clouds)
    assert L.hsp_rf_conv_bwd_scatter(one, one, one, null, one, one, 1, 8, 7, 128, one, one, null, 0, null) == -3


def test_state_dict_surface_matches_reference(flags, state_keys):
    """parameter names / shapes == the reference's (SURVEY 5 'checkpoint': 160 tensors train, 107 eval)."""
    from hs_pose_amd.PoseNet9D import PoseNet9D
    flags.train = 1
    sd = PoseNet9D().state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == state_keys["train"] and len(sd) == 160
    flags.train = 0
    sd = PoseNet9D().state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == state_keys["eval"] and len(sd) == 107
    assert sum(p.numel() for p in PoseNet9D().parameters()) < 9709871
    flags.train = 1
    assert sum(p.numel() for p in PoseNet9D().parameters()) == 9709871      # SURVEY 8c known answer


def test_layer_constructors_and_init(flags):
    """ctor signatures and init ranges of gcn3d.py:64-77, :117-141."""
    import math
    from hs_pose_amd import gcn3d
    s = gcn3d.HSlayer_surface(kernel_num=32, support_num=3)
    assert s.directions.shape == (3, 96) and s.STE_layer.weight.shape == (32, 3, 1) and s.conv2.weight.shape == (32, 64, 1)
    assert s.directions.abs().max().item() <= 1 / math.sqrt(96) + 1e-7
    h = gcn3d.HS_layer(in_channel=16, out_channel=32, support_num=3)
    assert h.weights.shape == (16, 128) and h.bias.shape == (128,) and h.directions.shape == (3, 96)
    bound = 1 / math.sqrt(32 * 4)
    for p in (h.weights, h.bias, h.directions):
        assert p.abs().max().item() <= bound + 1e-7
    pl = gcn3d.Pool_layer(pooling_rate=4, neighbor_num=4)
    assert pl.pooling_rate == 4 and pl.neighbor_num == 4 and len(list(pl.parameters())) == 0


def test_no_cpu_fallback(flags):
    """the product path fails loudly off-GPU instead of degrading to eager torch."""
    from hs_pose_amd import gcn3d, ops
    from hs_pose_amd._lib import HspError
    from hs_pose_amd.FaceRecon import FaceRecon
    with pytest.raises(HspError):
        gcn3d.get_neighbor_index(torch.zeros(1, 32, 3), 4)
    with pytest.raises(HspError):
        ops.gather_rows(torch.zeros(1, 8, 4), torch.zeros(1, 3, dtype=torch.int32))
    flags.train = 0
    with pytest.raises(HspError):
        FaceRecon()(torch.zeros(1, 64, 3), torch.zeros(1, 1))


def test_torch_binding_loads_and_rejects_cpu_tensors():
    """hs_pose_amd/_hsp_torch.so (csrc/hsp_torch.cpp): the reference extension's four entry points (chamfer_distance.cpp:180-185)
    and the inference calls are exported, resolve against the same libhsp.so, and refuse CPU tensors before any launch"""
    from hs_pose_amd._ext import available, ext
    assert available(), "run `make -C hs_pose_amd/csrc` (or __graft_entry__.build())"
    m = ext()
    for name in ("forward", "forward_cuda", "backward", "backward_cuda", "get_neighbor_index", "get_nearest_index", "knn_exact",
                 "hs_layer_forward", "surface_layer_forward", "pool_forward", "bn_eval", "center_cloud"):
        assert callable(getattr(m, name)), name
    x = torch.zeros(1, 8, 3)
    d = torch.zeros(1, 8)
    i = torch.zeros(1, 8, dtype=torch.int)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        m.forward_cuda(x, x, d, d, i, i)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        m.forward(x, x, d, d, i, i)                      # (the reference's CPU entry point has no counterpart: no CPU path)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        m.center_cloud(x)
    from hs_pose_amd.chamfer import ChamferDistance
    with pytest.raises(RuntimeError, match="GPU tensor"):
        ChamferDistance()(x, x)


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    from hs_pose_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.HspError, match="not built"):
        _lib.lib()


def test_product_never_imports_oracle():
    """hs_pose_amd/ must not reference oracle/ (the oracle is the checker, never the thing shipped)."""
    pkg = os.path.join(ROOT, "hs_pose_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "ref_cpu" not in txt and "hsp_oracle" not in txt.replace("oracle/hsp_oracle.c", ""), f

def test_round2_entry_points_validate_arguments_without_gpu():
    """the round-2 entry points (dense products, losses, augmentation, split weight-gradient fold) reject bad arguments
    before any launch -- checked here on the CPU"""
    from hs_pose_amd._lib import HspLossCfg, HspWgradPending, lib
    L = lib()
    null, one = ctypes.c_void_p(0), ctypes.c_void_p(64)
    f = ctypes.c_float
    # hsp_gemm_rows_f32(A1, lda1, B1, ldb1, layout1, K1, A2, lda2, B2, ldb2, layout2, K2, M, N, bias, resid, ldr, cbias, rpc, alpha,
    #                   xyz3, w3, C, ldc, ws, ws_bytes, stream)
    g = lambda *a: L.hsp_gemm_rows_f32(*a)
    assert g(null, 8, one, 8, 0, 8, null, 0, null, 0, 0, 0, 4, 4, null, null, 0, null, 0, f(1), null, null, one, 4, null, 0, null) == -1
    assert g(one, 4, one, 8, 0, 8, null, 0, null, 0, 0, 0, 4, 4, null, null, 0, null, 0, f(1), null, null, one, 4, null, 0, null) == -1  # lda < K
    assert g(one, 8, one, 8, 0, 8, null, 0, null, 0, 0, 0, 4, 4, null, null, 0, null, 0, f(1), null, null, one, 2, null, 0, null) == -1  # ldc < N
    assert g(one, 8, one, 8, 0, 8, null, 0, null, 0, 0, 0, 4, 4, null, null, 0, null, 0, f(1), one, null, one, 4, null, 0, null) == -1  # xyz3 w/o w3
    assert L.hsp_gemm_rows_workspace_bytes(0, 4, 4, 0, 4) == 0 and L.hsp_gemm_rows_workspace_bytes(1024, 256, 4608, 0, 4) > 0
    # bf16 form: "nn + nn" does not exist, elem size checked by the query
    assert L.hsp_gemm_rows_workspace_bytes(64, 64, 64, 0, 3) == 0
    # losses
    cfg = HspLossCfg()
    args17 = [one] * 17
    assert L.hsp_pose_losses_workspace_bytes(0) == 0 and L.hsp_pose_losses_workspace_bytes(16) > 0
    assert L.hsp_pose_losses_fwd(*([null] + [one] * 16), 2, 8, ctypes.byref(cfg), one, one, 1 << 20, null) == -1     # null input
    assert L.hsp_pose_losses_fwd(*args17, 2, 0, ctypes.byref(cfg), one, one, 1 << 20, null) == -1                    # N = 0
    assert L.hsp_pose_losses_fwd(*args17, 2, 8, ctypes.byref(cfg), one, null, 0, null) == -3                         # no workspace
    assert L.hsp_pose_losses_bwd(*args17, 2, 8, ctypes.byref(cfg), one, one, 16, one, *([one] * 10), null) == -3      # workspace too small
    assert L.hsp_pose_losses_bwd(*args17, 2, 8, ctypes.byref(cfg), null, one, 1 << 20, one, *([one] * 10), null) == -1  # no grad_terms
    # augmentation
    fl = [f(0.3)] * 4
    assert L.hsp_pose_augment(*([one] * 14), 2, 8, 0, *fl, one, one, one, one, null) == -1                            # M = 0
    assert L.hsp_pose_augment(*([one] * 13 + [null]), 2, 8, 4, *fl, one, one, one, one, null) == -1                   # no noise
    # split weight gradient: the fold checks its table
    pend = (HspWgradPending * 2)()
    assert L.hsp_wgrad_fold(pend, 0, null) == -1 and L.hsp_wgrad_fold(pend, 25, null) == -1
    assert L.hsp_wgrad_fold(pend, 1, null) == -1                                                                      # empty entry
    assert L.hsp_step_fold(null, 1, null, 0, null) == -1 and L.hsp_step_fold(pend, 1, null, 0, null) == -1           # no table / empty entry
    assert L.hsp_step_fold(pend, 0, null, 9, null) == -1 and L.hsp_step_fold(null, 0, null, 0, null) == 0            # nothing pending: no launch
    assert L.hsp_rf_conv_bwd_scatter_partial(one, one, one, null, one, one, 2, 64, 7, 64, one, one, one, 1 << 20, null, null) == -1   # no pending
    assert L.hsp_wgrad_partial_f32(one, 128, one, 1024, 128, 1024, 4000, one, 1024, null, one, 1 << 30, null, null) == -1   # no pending
