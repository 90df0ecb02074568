"""GPU: hsp_gemm_x3_f32 (fp32 products from exact three-way bf16 splits on the bf16 matrix cores, csrc/gemm_x3.hip) against
fp64 on the shapes of the layer path and of the heads: the error must be of the size an fp32 GEMM commits (it is compared with
torch.mm's own error on the same operands), for every epilogue, ragged row counts, split-K and an unaligned K."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref64(A1, B1, nn1, A2, B2, nn2, bias, resid, cb, rpc, alpha):
    r = A1.double() @ (B1.double() if nn1 else B1.double().t())
    if A2 is not None:
        r = r + A2.double() @ (B2.double() if nn2 else B2.double().t())
    r = alpha * r
    if bias is not None:
        r = r + bias.double()
    if resid is not None:
        r = r + resid.double()
    if cb is not None:
        r = r + cb.double().repeat_interleave(rpc, dim=0)[:r.shape[0]]
    return r


CASES = [  # M, N, K1, nn1, K2, nn2, epilogue, rows per cloud
    (16448, 1024, 128, True, 0, False, "bias", 0),            # fm = X W + b (conv_1)
    (4112, 2048, 256, True, 0, False, "bias", 0),             # fm (conv_3)
    (16448, 128, 128, False, 128, False, "rc", 1028),         # out = X Wste^T + F Wa^T + F + t[cloud]
    (4112, 256, 128, False, 256, False, "rc", 257),
    (16448, 128, 128, True, 0, False, "none", 0),             # gF = g Wa
    (16448, 128, 128, True, 1024, False, "none", 0),          # gX = g Wste + gfm W^T (conv_1): 257 tiles
    (4112, 256, 256, True, 2048, False, "none", 0),           # gX (conv_3): split-K
    (1024, 256, 512, True, 4096, False, "none", 0),           # gX (conv_4): split-K
    (9000, 128, 96, False, 0, False, "bias", 0),              # ragged rows, K not a multiple of 32 (k tail masked)
    (5777, 256, 40, True, 24, False, "none", 0),
    (16448, 1024, 1286, False, 0, False, "bias", 0),          # a head tower's first layer on feat rows (pitch 1288)
    (16448, 256, 1024, False, 0, False, "bias", 0),
    (16448, 1286, 1024, True, 0, False, "none", 0),           # its input gradient: N = 1286 (ragged last column tile)
    (4112, 200, 2048, False, 0, False, "none", 0),            # ragged N with split-K
    # the panel form (K = 128, weight fragments resident in registers: gemm_x3_panel_kernel) and its neighbours on the tile kernel
    (4112, 2048, 128, True, 0, False, "bias", 0),             # fm (conv_2): panel
    (4099, 1024, 128, False, 0, False, "none", 0),            # panel, ragged last row tile, alpha
    (49152, 1024, 128, True, 0, False, "bias", 0),            # panel, 24 tiles per workgroup
    (1024, 4096, 256, True, 0, False, "bias", 0),             # fm (conv_4): K = 256, tile kernel
    (8000, 1024, 128, False, 128, False, "none", 0),          # two sources: tile kernel
    (5001, 1024, 64, True, 64, False, "bias", 0),
]


@pytest.mark.parametrize("M,N,K1,nn1,K2,nn2,epi,rpc", CASES)
def test_gemm_x3_matches_fp64(dev, M, N, K1, nn1, K2, nn2, epi, rpc):
    from hs_pose_amd import ops
    g = torch.Generator().manual_seed(M + N + K1 + K2)
    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(dev)
    rowscale = (1.0 + 3.0 * torch.rand(M, 1, generator=g)).to(dev)         # uneven row magnitudes
    if K1 == 1286:                                    # rows of a wider, 16-byte pitched buffer (feat)
        full = rnd(M, 1288) * rowscale
        full[:, 1286:] = 0
        A1 = full[:, :1286]
    else:
        A1 = rnd(M, K1) * rowscale
    B1 = rnd(K1, N, scale=0.05) if nn1 else rnd(N, K1, scale=0.05)
    A2 = B2 = None
    if K2:
        A2 = rnd(M, K2)
        B2 = rnd(K2, N, scale=0.05) if nn2 else rnd(N, K2, scale=0.05)
    bias = rnd(N) if epi == "bias" else None
    resid = rnd(M, N) if epi == "rc" else None
    cb = rnd((M + rpc - 1) // rpc, N) if epi == "rc" else None
    alpha = 0.5 if epi == "none" else 1.0
    assert ops.gemm_x3_ok(A1, B1, A2, B2, bias, resid, cb, None, None, M, N)
    got = ops.gemm_x3(A1, B1, nn1, A2, B2, nn2, bias=bias, resid=resid, cloud_bias=cb, rows_per_cloud=rpc, alpha=alpha)
    want = _ref64(A1, B1, nn1, A2, B2, nn2, bias, resid, cb, rpc, alpha)
    # yardstick: the fp32 library product of the same operands
    lib32 = A1 @ (B1 if nn1 else B1.t())
    if K2:
        lib32 = lib32 + A2 @ (B2 if nn2 else B2.t())
    lib32 = alpha * lib32.double()
    if bias is not None:
        lib32 = lib32 + bias.double()
    if resid is not None:
        lib32 = lib32 + resid.double()
    if cb is not None:
        lib32 = lib32 + cb.double().repeat_interleave(rpc, dim=0)[:M]
    scale = want.abs().max().item()
    err = (got.double() - want).abs().max().item() / scale
    err_lib = (lib32 - want).abs().max().item() / scale
    rms = (got.double() - want).pow(2).mean().sqrt().item() / scale
    rms_lib = (lib32 - want).pow(2).mean().sqrt().item() / scale
    print(f"x3 M{M} N{N} K{K1}+{K2}: max err {err:.2e} of scale (fp32 library {err_lib:.2e}); rms {rms:.2e} (library {rms_lib:.2e})")
    # measured worst ratios (profiles/r04/x3_error.txt): 2.8x (max) and 1.7x (rms), both on the two-source out products
    assert err <= max(3.5 * err_lib, 2e-6), (err, err_lib)
    assert rms <= max(2.0 * rms_lib, 5e-7), (rms, rms_lib)


def test_gemm_x3_split_is_exact(dev):
    """the three slices add up to the fp32 value exactly (hi + mid + lo == x, in that order, in fp32)"""
    from hs_pose_amd import ops
    g = torch.Generator().manual_seed(3)
    W = (torch.randn(256, 96, generator=g) * torch.exp(4 * torch.randn(256, 96, generator=g))).to(dev)
    for transpose in (False, True):
        planes, kp, ps = ops.x3_planes.planes(W, transpose)
        N, K = (96, 256) if transpose else (256, 96)
        assert planes.shape == (3, N, kp) and kp % 32 == 0 and kp >= K
        hi, mid, lo = (planes[i, :, :K].float() for i in range(3))
        want = W.t() if transpose else W
        assert torch.equal((hi + mid) + lo, want)
        assert bool((planes[:, :, K:] == 0).all())


@pytest.mark.parametrize("M,N,K", [(16, 128, 128), (16, 512, 512), (16, 256, 256), (2, 128, 256), (5, 96, 40), (64, 128, 128), (40, 512, 256)])
def test_small_rows_and_outer(dev, M, N, K):
    """the per-cloud products of the ORL branch (one row per cloud): both weight layouts, a strided weight block, the outer
    product; against fp64"""
    from hs_pose_amd import ops
    g = torch.Generator().manual_seed(M * N + K)
    A = torch.randn(M, K, generator=g).to(dev)
    Wfull = (torch.randn(N, 2 * K, generator=g) * 0.1).to(dev)
    W = Wfull[:, K:]                                           # a column block of a wider (N, 2K) matrix (conv2's f_global half)
    got = ops.small_rows(A, W, False)
    want = A.double() @ W.double().t()
    assert (got.double() - want).abs().max().item() <= 2e-6 * max(1.0, want.abs().max().item())
    Wn = (torch.randn(K, 2 * N, generator=g) * 0.1).to(dev)[:, :N]
    got = ops.small_rows(A, Wn, True, alpha=0.25)
    want = 0.25 * (A.double() @ Wn.double())
    assert (got.double() - want).abs().max().item() <= 2e-6 * max(1.0, want.abs().max().item())
    c = torch.randn(M, N, generator=g).to(dev)
    out = torch.empty(K, 2 * N, device=dev)
    ops._tiny_tn(A, c, out[:, N:])
    want = A.double().t() @ c.double()
    assert (out[:, N:].double() - want).abs().max().item() <= 2e-6 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_colsum_rows_xyz(dev, dtype):
    """per-cloud column sums + coordinate moments of a gradient tensor in one pass, and the (C, 3) STE gradient summed over the
    batch as a rider of the outer-product launch (gcn3d.py:85), fp32 and bf16 rows, against fp64"""
    from hs_pose_amd import ops
    B, N, C = 5, 777, 128
    g_ = torch.Generator().manual_seed(11)
    g = torch.randn(B, N, C, generator=g_).to(dev).to(dtype)
    xyz = (torch.randn(B, N, 3, generator=g_) * 0.1).to(dev)
    mom = ops.colsum_rows_xyz(g, xyz)
    gd = g.double()
    want0 = gd.sum(dim=1)
    assert (mom[:, :C].double() - want0).abs().max().item() <= 1e-5 * want0.abs().max().item()
    for j in range(3):
        w = (gd * xyz[:, :, j:j + 1].double()).sum(dim=1)
        assert (mom[:, (1 + j) * C:(2 + j) * C].double() - w).abs().max().item() <= 1e-5 * max(1.0, w.abs().max().item())
    fg = torch.randn(B, C, generator=g_).to(dev)
    out = torch.empty(C, C, device=dev)
    gste = torch.empty(C, 3, device=dev)
    ops._tiny_tn(mom[:, :C], fg, out, mom=mom, gste=gste)
    want = gd.reshape(B * N, C).t() @ xyz.double().reshape(B * N, 3)
    assert (gste.double() - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())
    assert (out.double() - want0.t() @ fg.double()).abs().max().item() <= 1e-4 * max(1.0, (want0.t() @ fg.double()).abs().max().item())
