"""GPU parity of the hand-written dense per-point products (csrc/gemm_rows.hip) through the C-ABI against a plain
PyTorch reference of the same expression (fp64 accumulation as the truth).  Tolerances: fp32 kernel -- fp32 round-off
class, 2e-6 of sum|a||b| scale; bf16 kernel -- exact products, fp32 accumulation, ONE bf16 rounding of the result: 2^-8
relative to the result plus the accumulation term."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(A1, B1, nn1, A2, B2, nn2, bias, resid, cb, rpc):
    d = lambda t: t.double()
    y = d(A1) @ (d(B1) if nn1 else d(B1).t())
    mag = d(A1).abs() @ (d(B1).abs() if nn1 else d(B1).abs().t())
    if A2 is not None:
        y = y + d(A2) @ (d(B2) if nn2 else d(B2).t())
        mag = mag + d(A2).abs() @ (d(B2).abs() if nn2 else d(B2).abs().t())
    if bias is not None:
        y = y + d(bias)
    if resid is not None:
        y = y + d(resid)
    if cb is not None:
        rows = torch.arange(y.shape[0], device=y.device) // rpc
        y = y + d(cb)[rows]
    return y, mag


CASES = [
    # M, N, K1, nn1, K2, nn2, bias, resid, cloud rows (0 = none)
    (16448, 1024, 128, True, 0, False, True, False, 0),        # fm = X W + b (conv_1)
    (16448, 128, 128, False, 128, False, False, True, 1028),   # out = X Wste^T + F Wa^T + F + t[b]
    (16448, 128, 3, False, 128, False, False, True, 1028),     # conv_0: K1 = 3 (unaligned xyz rows)
    (4112, 2048, 256, True, 0, False, True, False, 0),         # conv_3 fm
    (4112, 256, 256, True, 2048, False, False, False, 0),      # gX = g Wste + gfm W^T ("nn" + "nt")
    (1024, 512, 256, False, 512, False, False, True, 64),      # conv_4 out: tiles span several clouds
    (300, 200, 77, False, 0, False, True, False, 0),           # ragged everything
    (257, 96, 40, True, 33, False, False, True, 100),          # nn + nt, ragged, unaligned (K2 = 33)
    (515, 130, 6, False, 34, True, True, False, 0),            # nt + nn (issued with the sources swapped), 8-byte rows
    (2056, 1024, 1286, False, 0, False, True, False, 0),       # a head's first layer: K = 1286 (8-byte aligned rows)
    (2056, 1286, 1024, True, 0, False, False, False, 0),       # its input gradient: N = 1286
    (2056, 30, 128, False, 0, False, True, False, 0),          # face_head's last layer: N = 30
    (64, 64, 32, False, 0, False, False, False, 0),            # one small tile, one k block
    (1024, 256, 512, True, 4096, False, False, False, 0),      # conv_4's input gradient: 64 tiles x K = 4608 -> split-K
    (1024, 128, 2000, False, 0, False, True, True, 100),       # split-K with the whole epilogue (bias, resid, cloud bias)
    (16, 128, 128, False, 0, False, False, False, 0),          # the per-cloud products (16 rows)
    # full-size rows (16448 = 257 x 64: whole tiles fill 256 CUs unevenly)
    (16448, 1024, 1286, False, 0, False, True, False, 0),      # a head's first layer at full size
    (16448, 128, 128, True, 1024, False, False, True, 1028),   # gX = g Wste + gfm W^T: 514 tiles of 64 x 64, residual + cloud bias
    (16448, 256, 1024, False, 0, False, True, True, 0),        # 16448 x 1024 -> 256
]


@pytest.mark.parametrize("M,N,K1,nn1,K2,nn2,use_bias,use_resid,rpc", CASES)
def test_gemm_rows_f32(dev, ref, M, N, K1, nn1, K2, nn2, use_bias, use_resid, rpc):
    from hs_pose_amd import ops
    h = lambda shape, seed: ref.hash_tensor(shape, seed, 1.0).to(dev)
    A1 = h((M, K1), 1)
    B1 = h((K1, N) if nn1 else (N, K1), 2)
    A2 = h((M, K2), 3) if K2 else None
    B2 = (h((K2, N) if nn2 else (N, K2), 4)) if K2 else None
    bias = h((N,), 5) if use_bias else None
    resid = h((M, N), 6) if use_resid else None
    cb = h(((M + rpc - 1) // rpc, N), 7) if rpc else None
    got = ops.gemm_rows(A1, B1, nn1, A2, B2, nn2, bias, resid, cb, rpc)
    want, mag = _ref(A1, B1, nn1, A2, B2, nn2, bias, resid, cb, rpc)
    err = ((got.double() - want).abs() / (mag + 1.0)).max().item()
    assert err <= 2e-6, f"max error {err:.3e} of the |a||b| scale"


def test_gemm_rows_f32_strided_views(dev, ref):
    """operands / outputs that are column blocks of wider tensors (conv2 = [Wa | Wb], gradient blocks)"""
    from hs_pose_amd import ops
    h = lambda shape, seed: ref.hash_tensor(shape, seed, 1.0).to(dev)
    g, conv2 = h((1000, 128), 11), h((128, 256), 12)
    Wa = conv2[:, :128]
    got = ops.gemm_rows(g, Wa, nn1=True)                     # gF = g Wa : "nn" with ldb = 2C
    assert torch.allclose(got, g @ Wa, atol=2e-4, rtol=1e-5)
    F = h((1000, 128), 13)
    wide = torch.zeros(1000, 300, device=dev)
    ops.gemm_rows(F, Wa, out=wide[:, 100:228])               # F Wa^T into a column block
    assert torch.allclose(wide[:, 100:228], F @ Wa.t(), atol=2e-4, rtol=1e-5)
    assert float(wide[:, :100].abs().max()) == 0.0 and float(wide[:, 228:].abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K1,K2,use_bias,use_resid,rpc", [
    (8192, 1024, 128, 0, True, False, 0), (8192, 128, 128, 128, False, True, 4096), (4096, 2048, 256, 0, True, False, 0),
    (1000, 136, 72, 40, True, True, 300), (2048, 1024, 1288, 0, True, False, 0), (64, 64, 64, 0, False, False, 0),
    # enough tiles for the 256 x 128 workgroup tile (128 x 64 per wave): plain, ragged rows, dual source + every epilogue
    (65536, 512, 128, 0, True, False, 0), (65500, 520, 200, 0, False, False, 0), (65536, 512, 128, 128, True, True, 4096),
    (16448, 1024, 1288, 0, True, False, 0), (16448, 128, 1024, 128, False, True, 1028)])
def test_gemm_rows_bf16(dev, ref, M, N, K1, K2, use_bias, use_resid, rpc):
    from hs_pose_amd import ops
    h = lambda shape, seed: ref.hash_tensor(shape, seed, 1.0).to(dev)
    A1, B1 = h((M, K1), 1).bfloat16(), h((N, K1), 2).bfloat16()
    A2 = h((M, K2), 3).bfloat16() if K2 else None
    B2 = h((N, K2), 4).bfloat16() if K2 else None
    bias = h((N,), 5) if use_bias else None
    resid = h((M, N), 6).bfloat16() if use_resid else None
    cb = h(((M + rpc - 1) // rpc, N), 7) if rpc else None
    got = ops.gemm_rows(A1, B1, False, A2, B2, False, bias, resid, cb, rpc)
    assert got.dtype == torch.bfloat16
    want, mag = _ref(A1, B1, False, A2, B2, False, bias, resid, cb, rpc)
    # one bf16 rounding of the result (2^-9 relative, half an ulp) + fp32 accumulation of exact products
    tol = want.abs() * 2.0 ** -8 + 4e-6 * (mag + 1.0)
    bad = ((got.double() - want).abs() > tol).sum().item()
    assert bad == 0, f"{bad} entries beyond one bf16 rounding"


# ---- the LDS-free wave-level kernel (csrc/gemm_wave.hip): every instantiated form x tile shape, the leftover-block path, strided
# operands; same truth and tolerance as gemm_rows
WAVE_CASES = [
    # M, N, K1, nn1, K2, nn2, bias, resid+cloud rows (0 = none), xyz3
    (16448, 1024, 128, True, 0, False, True, 0, False),        # fm = X W + b (conv_1): 32 x 128 tiles, leftover blocks
    (4112, 2048, 256, True, 0, False, True, 0, False),         # conv_3 fm: 32 x 64 tiles
    (1024, 4096, 256, True, 0, False, True, 0, False),         # conv_4 fm
    (16448, 128, 128, True, 0, False, False, 0, False),        # g Wa
    (1024, 512, 512, True, 0, False, False, 0, False),         # g Wa at the coarsest level: 32 x 32 tiles
    (16448, 128, 128, False, 128, False, False, 1028, False),  # out = X Wste^T + F Wa^T + F + t[b]
    (4112, 256, 256, False, 256, False, False, 257, False),    # conv_3 out
    (16448, 128, 128, False, 0, False, False, 1028, True),     # conv_0 out: the K = 3 STE rides in the epilogue
    (16448, 128, 128, True, 1024, False, False, 0, False),     # gX = g Wste + gfm W^T ("nn" + "nt")
    (4112, 256, 128, False, 0, False, False, 0, False),        # x W^T
    (4112, 256, 128, False, 0, False, True, 0, False),         # x W^T + b
    (1000, 96, 64, False, 0, False, True, 0, False),           # ragged rows, N = 3 x 32
    (33, 32, 32, True, 0, False, False, 0, False),             # two row blocks, one of them a single row
]


@pytest.mark.parametrize("cfg", [0, 0x10042, 0x20041, 0x20021, 0x20011, 0x10020041])
@pytest.mark.parametrize("M,N,K1,nn1,K2,nn2,use_bias,rpc,use_xyz", WAVE_CASES)
def test_gemm_wave_f32(dev, ref, M, N, K1, nn1, K2, nn2, use_bias, rpc, use_xyz, cfg):
    from hs_pose_amd import ops
    if not ops.gemm_wave_supported(M, N, K1, K2, cfg):
        pytest.skip("tile shape does not divide N")
    h = lambda shape, seed: ref.hash_tensor(shape, seed, 1.0).to(dev)
    A1 = h((M, K1), 1)
    B1 = h((K1, N) if nn1 else (N, K1), 2)
    A2 = h((M, K2), 3) if K2 else None
    B2 = (h((K2, N) if nn2 else (N, K2), 4)) if K2 else None
    bias = h((N,), 5) if use_bias else None
    resid = h((M, N), 6) if rpc else None
    cb = h(((M + rpc - 1) // rpc, N), 7) if rpc else None
    xyz3, w3 = (h((M, 3), 8), h((N, 3), 9)) if use_xyz else (None, None)
    out = torch.full((M, N), float("nan"), device=dev)
    ops.gemm_wave(A1, B1, nn1, A2, B2, nn2, bias, resid, cb, rpc, out=out, xyz3=xyz3, w3=w3, cfg=cfg)
    want, mag = _ref(A1, B1, nn1, A2, B2, nn2, bias, resid, cb, rpc)
    if use_xyz:
        want = want + xyz3.double() @ w3.double().t()
        mag = mag + xyz3.double().abs() @ w3.double().abs().t()
    err = ((out.double() - want).abs() / (mag + 1.0)).max().item()
    assert err <= 2e-6, f"max error {err:.3e} of the |a||b| scale"


def test_gemm_wave_strided_and_unsupported(dev, ref):
    """column blocks of wider tensors as operands / output; shapes and forms outside the kernel's cover are refused (the host
    routes them to gemm_rows)"""
    from hs_pose_amd import ops
    from hs_pose_amd._lib import HspError
    h = lambda shape, seed: ref.hash_tensor(shape, seed, 1.0).to(dev)
    g, conv2 = h((4112, 256), 11), h((256, 512), 12)
    Wa = conv2[:, :256]
    got = ops.gemm_wave(g, Wa, nn1=True)                     # gF = g Wa : "nn" with ldb = 2C
    assert torch.allclose(got, g @ Wa, atol=3e-4, rtol=1e-5)
    wide = torch.zeros(4112, 768, device=dev)
    ops.gemm_wave(g, Wa, out=wide[:, 256:512])               # g Wa^T into a column block
    assert torch.allclose(wide[:, 256:512], g @ Wa.t(), atol=3e-4, rtol=1e-5)
    assert float(wide[:, :256].abs().max()) == 0.0 and float(wide[:, 512:].abs().max()) == 0.0
    assert not ops.gemm_wave_supported(1000, 100, 64, 0)     # N not a multiple of 32
    assert not ops.gemm_wave_supported(1000, 128, 48, 0)     # K not a multiple of 32
    with pytest.raises(HspError):                            # bias + residual: not an instantiated form
        ops.gemm_wave(g, Wa, nn1=True, bias=h((256,), 13), resid=h((4112, 256), 14), cloud_bias=h((16, 256), 15), rows_per_cloud=257)
    # gemm_own picks a kernel that covers the call either way
    y = ops.gemm_own(g, Wa, True, bias=h((256,), 13), resid=h((4112, 256), 14), cloud_bias=h((16, 256), 15), rows_per_cloud=257)
    want = g @ Wa + h((256,), 13) + h((4112, 256), 14) + h((16, 256), 15).repeat_interleave(257, 0)
    assert torch.allclose(y, want, atol=3e-4, rtol=1e-5)
