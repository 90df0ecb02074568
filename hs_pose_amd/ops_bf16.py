"""bf16 feature storage for the HS stack (BASELINE.json configs[3]: dense clouds, "KNN LDS tiling + MFMA MLP path").

What is bf16: the layer inputs / outputs, ``fm = X W + b`` and the winners' support values ``fwin``, every activation
gradient, and the GEMM operands (working copies of the weights, made once per step from the fp32 master parameters by
ONE ``hsp_cast_params_bf16`` launch).  What stays fp32: xyz, the support directions and theta -- the reference hard-casts
the receptive field with ``.float()`` (network/fs_net_repo/gcn3d.py:57,59) --, neighbour indices / arg-max, BatchNorm
statistics and affine parameters, the per-cloud ORL rows (B,C), every parameter gradient and the master parameters
themselves (state_dict unchanged).  Every kernel accumulates in fp32.

The two layer nodes below mirror ``ops._HSLayer`` / ``ops._SurfaceLayer`` kernel for kernel:
    forward   fm = X W + b                      hsp_gemm_rows_bf16 (bf16 MFMA, bias in the epilogue)
              F  = graph_conv(fm)               hsp_rf_conv_fwd_bf16
              fg = mean_i max_n F[idx_xyz]      hsp_orl_global_fwd_bf16 -> fp32 (B,C);  t = fg Wb^T in fp32
              out = X Wste^T + F Wa^T + F + t   hsp_gemm_rows_bf16, two sources, residual + per-cloud bias epilogue
                    (surface layer: the K = 3 STE on raw fp32 coordinates rides in the same epilogue)
    backward  the same kernels in their *_bf16 forms; parameter gradients from hsp_wgrad_bf16 in fp32.
"""
import ctypes
import weakref

import numpy as np
import torch

from . import ops
from ._lib import HspError, lib
from .ops import _p, _req, _run, _stream, _ws

_vp = ctypes.c_void_p
BF16 = torch.bfloat16

# parameter (by storage pointer) -> (bf16 copy, bf16 transposed copy, weak reference to the owning Bf16Params)
_copies = {}


def copies_of(param):
    hit = _copies.get(param.data_ptr())
    if hit is not None:                       # the address may have been re-used by another tensor since: the shape must match
        c, ct = hit[0], hit[1]
        shape = tuple(param.shape)
        if (c is not None and tuple(c.shape) != shape) or (ct is not None and tuple(ct.shape) != shape[::-1]):
            hit = None
    if hit is None:
        raise HspError("bf16 path: no bf16 working copy registered for this parameter (FaceRecon.set_feature_dtype)")
    return hit[0], hit[1]


class Bf16Params:
    """bf16 working copies (and (N,K)-form transposes) of the GEMM weights of a module tree, refreshed by one launch.

    ``specs``: list of (parameter viewed as a 2-D fp32 matrix, want_copy, want_transposed)."""

    def __init__(self, specs):
        dev = specs[0][0].device
        for key in [k_ for k_, v_ in _copies.items() if v_[2] is not None and v_[2]() is None]:
            del _copies[key]                  # entries whose Bf16Params is gone
        self.entries = []
        tab = np.zeros(len(specs), dtype=np.dtype([("src", np.uint64), ("dst", np.uint64), ("dstT", np.uint64),
                                                   ("rows", np.int32), ("cols", np.int32), ("ld", np.int32),
                                                   ("tile0", np.int32)]))
        tiles = 0
        for i, (w2, want, want_t) in enumerate(specs):
            if w2.dim() != 2 or w2.dtype != torch.float32 or w2.stride(1) != 1:
                raise HspError("Bf16Params: fp32 matrices with contiguous rows")
            rows, cols = w2.shape
            c = torch.empty(rows, cols, dtype=BF16, device=dev) if want else None
            ct = torch.empty(cols, rows, dtype=BF16, device=dev) if want_t else None
            self.entries.append((w2, c, ct))
            _copies[w2.data_ptr()] = (c, ct, weakref.ref(self))
            tab[i] = (w2.data_ptr(), c.data_ptr() if c is not None else 0, ct.data_ptr() if ct is not None else 0, rows, cols,
                      w2.stride(0), tiles)
            tiles += ((rows + 31) // 32) * ((cols + 31) // 32)
        self.total_tiles = tiles
        self.n = len(specs)
        self.table = torch.from_numpy(tab.view(np.uint8)).to(dev)
        self.ptrs = [w2.data_ptr() for w2, _, _ in self.entries]

    def refresh(self):
        """round the current fp32 master weights into the working copies (call at the top of every forward)"""
        for (w2, _, _), ptr in zip(self.entries, self.ptrs):
            if w2.data_ptr() != ptr:
                raise HspError("Bf16Params: a parameter was re-seated (e.g. by building the fused optimizer); rebuild the "
                               "bf16 copies with FaceRecon.set_feature_dtype(torch.bfloat16)")
        _run("hsp_cast_params_bf16", (_p(self.table), self.n, self.total_tiles, _stream()), key=f"n{self.n}",
             abytes=6 * sum(w.numel() for w, _, _ in self.entries))


def _b(t, name):
    return _req(t, BF16, name)


def _orl_fwd(F3, idx_x, k):
    B, N, C = F3.shape
    fg = torch.empty(B, C, dtype=torch.float32, device=F3.device)
    arg = torch.empty(B, N, C, dtype=torch.uint8, device=F3.device)
    wsb = lib().hsp_orl_workspace_bytes(B, N, C)
    ws = _ws(wsb, F3.device)
    _run("hsp_orl_global_fwd_bf16", (_p(F3), _p(idx_x), B, N, k, idx_x.shape[2], C, _p(fg), _p(arg), _p(ws), wsb, _stream()),
         key=f"B{B}N{N}k{k}C{C}", abytes=B * N * (2 * C + 4 * k + C))
    return fg, arg


def _colsum(x3):
    B, N, C = x3.shape
    out = torch.empty(B, C, dtype=torch.float32, device=x3.device)
    wsb = lib().hsp_orl_workspace_bytes(B, N, C)
    ws = _ws(wsb, x3.device)
    _run("hsp_colsum_rows_bf16", (_p(x3), B, N, C, _p(out), _p(ws), wsb, _stream()), key=f"B{B}N{N}C{C}", abytes=2 * B * N * C)
    return out


def _wgrad(A2, B2, out=None, colsum=False):
    """fp32 (M,N) = A2^T B2 for bf16 point rows A2 (K,M), B2 (K,N) (+ fp32 column sums of B2)"""
    K, M = A2.shape
    N = B2.shape[1]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A2.device)
    if M % 64 or N % 64 or A2.stride(1) != 1 or B2.stride(1) != 1 or A2.stride(0) % 2 or B2.stride(0) % 2:
        raise HspError("bf16 weight gradient: channel counts must be multiples of 64")
    return ops._wgrad_custom(A2, B2, out, colsum)          # (hsp_wgrad_bf16, or its partial form inside an ops.WgradBatch)


def _orl_bwd_accumulate(gfg_over_n, idx_x, arg, gF3, extra):
    """gF3 += extra + gfg_over_n[b] * #{queries whose neighbourhood max came from this row}   (in place, one pass)"""
    B, N, C = gF3.shape
    _run("hsp_gather_max_bwd_bf16", (_p(gfg_over_n), 1, _p(idx_x), _vp(0), _p(arg), B, N, N, N, idx_x.shape[2], C, _p(gF3), 1,
                                     _p(extra), _stream()),
         key=f"B{B}Ns{N}Nq{N}C{C}bc+", abytes=B * N * (6 * C + 4 * idx_x.shape[2] + C))


def _rf_conv_fwd(xyz, idx, directions, fm3, S, need_bwd):
    B, N, k = idx.shape
    SC = directions.shape[1]
    C = SC // S
    out = torch.empty(B, N, C, dtype=BF16, device=xyz.device)
    arg = torch.empty(B, N, SC, dtype=torch.uint16, device=xyz.device)
    want = need_bwd and lib().hsp_rf_conv_wants_fwin_bf16(N, S, C)
    fwin = torch.empty(B, N, SC, dtype=BF16, device=xyz.device) if want else None
    _run("hsp_rf_conv_fwd_bf16", (_p(xyz), _p(idx), _p(directions), _p(fm3), B, N, k, S, C, _p(out), _p(arg), _p(fwin), _stream()),
         key=f"B{B}N{N}k{k}S{S}C{C}",
         abytes=B * N * (12 + 4 * k + 2 * (S + 1) * C + 2 * C + 2 * SC + (2 * SC if fwin is not None else 0)) + 12 * SC)
    return out, arg, fwin


def _rf_conv_bwd(xyz, directions, fm, arg, gF3, S):
    B, N, C = gF3.shape
    SC = directions.shape[1]
    gfm = torch.empty(B, N, (S + 1) * C, dtype=BF16, device=gF3.device)
    gd = torch.empty_like(directions)
    wsb = lib().hsp_rf_bwd_scatter_workspace_bytes(B, SC)
    ws = _ws(wsb, gF3.device)
    is_fwin = fm.shape[-1] == SC
    ops._rf_bwd_dirs_call("hsp_rf_conv_bwd_scatter_bf16", (_p(xyz), _p(directions), _p(None if is_fwin else fm),
                                                           _p(fm if is_fwin else None), _p(arg), _p(gF3), B, N, S, C, _p(gfm), _p(gd)),
                          ws, wsb, (directions, gd), key=f"B{B}N{N}S{S}C{C}",
                          abytes=B * N * (12 + 2 * SC + 2 * SC + 2 * C + 2 * (S + 1) * C) + 24 * SC)
    return gfm, gd


class _HSLayerBf16(torch.autograd.Function):
    """HS_layer.forward (gcn3d.py:143-156) on bf16 feature rows; parameters are the fp32 masters (gradients fp32)."""

    @staticmethod
    def forward(ctx, xyz, X, idx_f, idx_x, k, S, weights, bias, directions, w_ste3, w_conv23, out_f32):
        w_ste, w_conv2 = w_ste3.squeeze(-1), w_conv23.squeeze(-1)
        xyz = _req(xyz, torch.float32, "hs_layer.xyz")
        X = _b(X, "hs_layer.X")
        idx_f = _req(idx_f, torch.int32, "hs_layer.idx_f")
        idx_x = _req(idx_x, torch.int32, "hs_layer.idx_x")
        directions = _req(directions, torch.float32, "hs_layer.directions")
        B, N, Cin = X.shape
        SC = directions.shape[1]
        C = SC // S
        W_b, WT_b = copies_of(weights)                           # (Cin,(S+1)C) and its (N,K) form ((S+1)C, Cin)
        ste_b, _ = copies_of(w_ste)                               # (C, Cin)
        c2_b, _ = copies_of(w_conv2)                              # (C, 2C)
        X2 = X.view(B * N, Cin)
        fm = ops.gemm_rows(X2, WT_b, bias=bias)                   # (BN,(S+1)C) bf16
        need_bwd = any(ctx.needs_input_grad)
        F3, arg, fwin = _rf_conv_fwd(xyz, idx_f, directions, fm.view(B, N, -1), S, need_bwd)
        if fwin is not None:
            fm = fwin
        fm = fm.view(B, N, -1)
        fg, arg_o = _orl_fwd(F3, idx_x, k)                        # fp32 (B,C)
        t2 = ops._mm_nt(fg, w_conv2[:, C:])                       # fp32 per-cloud half of conv2
        F2 = F3.view(B * N, C)
        # a layer output that feeds BatchNorm is written in fp32 (its per-channel mean is often >> its std: bf16 would leave
        # the normalised value a handful of significant bits); the gradient that comes back is bf16 either way
        out3 = torch.empty(B, N, C, dtype=torch.float32 if out_f32 else BF16, device=X.device)
        ops.gemm_rows(X2, ste_b, False, F2, c2_b[:, :C], False, resid=F2, cloud_bias=t2.contiguous(), rows_per_cloud=N,
                      out=out3.view(B * N, C))
        ctx.save_for_backward(xyz, X, idx_f, idx_x, fm, arg, F3, arg_o, fg, weights, directions, w_ste3, w_conv23)
        ctx.k, ctx.S = k, S
        return out3

    @staticmethod
    def backward(ctx, g):
        xyz, X, idx_f, idx_x, fm, arg, F3, arg_o, fg, weights, directions, w_ste3, w_conv23 = ctx.saved_tensors
        w_ste, w_conv2 = w_ste3.squeeze(-1), w_conv23.squeeze(-1)
        k, S = ctx.k, ctx.S
        if g.dtype == torch.float32:             # fp32 output (ahead of a BatchNorm): autograd hands its gradient back in fp32
            g = g.bfloat16()
        g = _b(g, "hs_layer.grad")
        B, N, Cin = X.shape
        C = F3.shape[2]
        W_b, _ = copies_of(weights)
        _, steT_b = copies_of(w_ste)                              # (Cin, C)
        _, c2T_b = copies_of(w_conv2)                             # (2C, C): rows [0,C) = Wa^T
        g2, X2, F2 = g.view(B * N, C), X.view(B * N, Cin), F3.view(B * N, C)
        Wb = w_conv2[:, C:]
        gt = _colsum(g)                                           # fp32 (B,C)
        g_conv2 = torch.empty_like(w_conv2)
        with ops.WgradBatch():                                        # the three parameter gradients: one fold launch
            _wgrad(g2, F2, out=g_conv2[:, :C])                        # gWa
            ops._tiny_tn(gt, fg, g_conv2[:, C:])                      # gWb (fp32, tiny)
            gF3 = torch.empty(B, N, C, dtype=BF16, device=g.device)
            ops.gemm_rows(g2, c2T_b[:C], out=gF3.view(B * N, C))      # g Wa ...
            _orl_bwd_accumulate(ops._mm_nn(gt, Wb, alpha=1.0 / N), idx_x, arg_o, gF3, g)      # ... + g + ORL scatter
            gfm, gD = _rf_conv_bwd(xyz, directions, fm.view(B, N, -1), arg, gF3, S)
            gfm2 = gfm.view(B * N, -1)
            gW, gb = _wgrad(X2, gfm2, colsum=True)
            g_ste = _wgrad(g2, X2)
            gX3 = torch.empty(B, N, Cin, dtype=BF16, device=g.device)
            ops.gemm_rows(g2, steT_b, False, gfm2, W_b, False, out=gX3.view(B * N, Cin))      # g Wste + gfm W^T
        return None, gX3, None, None, None, None, gW, gb, gD, g_ste.unsqueeze_(-1), g_conv2.unsqueeze_(-1), None


class _SurfaceLayerBf16(torch.autograd.Function):
    """HSlayer_surface.forward (gcn3d.py:79-90) producing bf16 rows; xyz carries no gradient."""

    @staticmethod
    def forward(ctx, xyz, idx_x, k, S, directions, w_ste3, w_conv23):
        w_ste, w_conv2 = w_ste3.squeeze(-1), w_conv23.squeeze(-1)
        xyz = _req(xyz, torch.float32, "surface_layer.xyz")
        idx_x = _req(idx_x, torch.int32, "surface_layer.idx")
        directions = _req(directions, torch.float32, "surface_layer.directions")
        B, N, _ = xyz.shape
        SC = directions.shape[1]
        C = SC // S
        if idx_x.shape[2] != k:
            raise HspError("surface_layer: idx must have exactly k columns")
        c2_b, _ = copies_of(w_conv2)
        F3 = torch.empty(B, N, C, dtype=BF16, device=xyz.device)
        arg = torch.empty(B, N, SC, dtype=torch.uint16, device=xyz.device)
        _run("hsp_rf_surface_fwd_bf16", (_p(xyz), _p(idx_x), _p(directions), B, N, k, S, C, _p(F3), _p(arg), _stream()),
             key=f"B{B}N{N}k{k}S{S}C{C}", abytes=B * N * (12 + 4 * k + 2 * C + 2 * SC) + 12 * SC)
        fg, arg_o = _orl_fwd(F3, idx_x, k)
        F2, x2 = F3.view(B * N, C), xyz.view(B * N, 3)
        t2 = ops._mm_nt(fg, w_conv2[:, C:])
        out3 = torch.empty(B, N, C, dtype=BF16, device=xyz.device)
        # F Wa^T + F + t[b] on the bf16 matrix cores; the K = 3 STE on the raw fp32 coordinates in the epilogue
        ops.gemm_rows(F2, c2_b[:, :C], False, resid=F2, cloud_bias=t2.contiguous(), rows_per_cloud=N, out=out3.view(B * N, C),
                      xyz3=x2, w3=w_ste.contiguous())
        ctx.save_for_backward(xyz, idx_x, arg, F3, arg_o, fg, directions, w_conv23)
        ctx.k, ctx.S = k, S
        return out3

    @staticmethod
    def backward(ctx, g):
        xyz, idx_x, arg, F3, arg_o, fg, directions, w_conv23 = ctx.saved_tensors
        w_conv2 = w_conv23.squeeze(-1)
        k, S = ctx.k, ctx.S
        g = _b(g, "surface_layer.grad")
        B, N, C = F3.shape
        SC = directions.shape[1]
        _, c2T_b = copies_of(w_conv2)
        g2, F2, x2 = g.view(B * N, C), F3.view(B * N, C), xyz.view(B * N, 3)
        Wb = w_conv2[:, C:]
        own_ste = ops._ste_moments_ok(C) and B <= 64
        g_conv2 = torch.empty_like(w_conv2)
        if own_ste:      # gt and the coordinate moments of g in one pass; g^T xyz = their sum over the batch (no cast, no GEMM)
            mom = ops.colsum_rows_xyz(g, xyz)
            gt = mom[:, :C]
            g_ste = torch.empty(C, 3, dtype=torch.float32, device=g.device)
        else:
            gt = _colsum(g)
        _wgrad(g2, F2, out=g_conv2[:, :C])
        if own_ste:
            ops._tiny_tn(gt, fg, g_conv2[:, C:], mom=mom, gste=g_ste)
        else:
            ops._tiny_tn(gt, fg, g_conv2[:, C:])
        gF3 = torch.empty(B, N, C, dtype=BF16, device=g.device)
        ops.gemm_rows(g2, c2T_b[:C], out=gF3.view(B * N, C))
        _orl_bwd_accumulate(ops._mm_nn(gt, Wb, alpha=1.0 / N), idx_x, arg_o, gF3, g)
        gD = torch.empty_like(directions)
        wsb = lib().hsp_rf_bwd_scatter_workspace_bytes(B, SC)
        ws = _ws(wsb, g.device)
        ops._rf_bwd_dirs_call("hsp_rf_surface_bwd_bf16", (_p(xyz), _p(directions), _p(arg), _p(gF3), B, N, S, C, _p(gD)), ws, wsb,
                              (directions, gD), key=f"B{B}N{N}S{S}C{C}", abytes=B * N * (12 + 2 * C + 2 * SC) + 24 * SC)
        if not own_ste:
            g_ste = g2.float().t() @ x2                           # (C,3): three columns -- not a matrix-core shape
        return None, None, None, None, gD, g_ste.unsqueeze_(-1), g_conv2.unsqueeze_(-1)


def hs_layer(xyz, X, idx_f, idx_x, k, S, weights, bias, directions, w_ste, w_conv2, out_f32=False):
    return _HSLayerBf16.apply(xyz, X, idx_f, idx_x, k, S, weights, bias, directions, w_ste, w_conv2, bool(out_f32))


def surface_layer(xyz, idx_x, k, S, directions, w_ste, w_conv2):
    return _SurfaceLayerBf16.apply(xyz, idx_x, k, S, directions, w_ste, w_conv2)
