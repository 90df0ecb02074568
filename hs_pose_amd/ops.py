"""Device ops of the hot path: thin autograd wrappers over the libhsp.so C-ABI (include/hsp.h).

Each function checks device / dtype / contiguity, allocates outputs with torch (PyTorch owns every
buffer), passes raw device pointers + the current HIP stream through ctypes and raises on a non-zero
return code.  No CPU fallback: a non-GPU tensor is an error.
"""
import ctypes
import os

import torch

from ._lib import HspError, check, lib

_vp = ctypes.c_void_p

# HSP_DETERMINISTIC=1 (or ops.DETERMINISTIC = True): graph-conv backward runs in gather form over a
# reverse-edge index (fixed summation order, bit-reproducible gradients) instead of the faster
# column-tile LDS scatter, whose fp32 LDS adds are order-dependent in the last bits.
DETERMINISTIC = os.environ.get("HSP_DETERMINISTIC", "0") == "1"


def _p(t):
    return _vp(t.data_ptr()) if t is not None else _vp(0)


def _stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


def _req(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise HspError(f"{name}: expected a GPU tensor (hs_pose_amd has no CPU path), got "
                       f"{getattr(t, 'device', type(t))}")
    if t.dtype != dtype:
        raise HspError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


_FEAT_DTYPES = (torch.float32, torch.bfloat16)


def _reqf(t, name, like=None):
    """a FEATURE tensor: fp32, or bf16 storage (BASELINE configs[3]; the kernels still compute in fp32).  ``like``: a
    tensor whose dtype it must share."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise HspError(f"{name}: expected a GPU tensor (hs_pose_amd has no CPU path), got "
                       f"{getattr(t, 'device', type(t))}")
    if t.dtype not in _FEAT_DTYPES or (like is not None and t.dtype != like.dtype):
        raise HspError(f"{name}: expected dtype {like.dtype if like is not None else 'float32 or bfloat16'}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _sfx(t):
    """entry-point suffix for a feature tensor's storage type"""
    return "_bf16" if t.dtype == torch.bfloat16 else ""


def _es(t):
    return 2 if t.dtype == torch.bfloat16 else 4


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


class KernelTimer:
    """Optional per-call timing of the C-ABI entry points with HIP events recorded on the stream the
    kernels are launched on (torch's current stream).  bench.py installs one over its timed region to
    get the live average duration of the dominant kernel for the roofline line; ``abytes`` is the
    call's ALGORITHMIC byte count (every input read once, every output written once -- DESIGN.md)."""

    def __init__(self, only=None):
        self.only = set(only) if only else None
        self.records = []          # (name, key, abytes, aflops, ev0, ev1)

    def summary(self):
        """{(name, key): dict(calls, total_ms, avg_us, abytes)} -- call after torch.cuda.synchronize()."""
        out = {}
        for name, key, ab, fl, e0, e1 in self.records:
            d = out.setdefault((name, key), dict(calls=0, total_ms=0.0, abytes=ab, aflops=fl))
            d["calls"] += 1
            d["total_ms"] += e0.elapsed_time(e1)
        for d in out.values():
            d["avg_us"] = 1e3 * d["total_ms"] / d["calls"]
        return out


_timer = None
# (entry point, key) -> bytes the kernel streams BY DESIGN where that differs from the algorithmic count of DESIGN.md section 4
# (the receptive-field kernels keep a uint16 winning-row slot instead of a uint8 neighbour slot and, for large clouds, a private
# stream of the winners' support values): bench.py reports both
design_stream_bytes = {}


def set_timer(t):
    """install (or with None remove) a KernelTimer; returns the previous one."""
    global _timer
    prev, _timer = _timer, t
    return prev


def _run(name, args, key="", abytes=0, aflops=0):
    """call libhsp entry point ``name``; raise on a non-zero return code.  abytes / aflops: the call's
    algorithmic bytes and (for GEMM-shaped, MFMA-bound kernels) flops, for bench.py's roofline line."""
    fn = getattr(lib(), name)
    t = _timer
    if t is not None and (t.only is None or name in t.only):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args)
        e1.record()
        t.records.append((name, key, abytes, aflops, e0, e1))
    else:
        rc = fn(*args)
    check(rc, name)


# ------------------------------------------------------------------------------------------------
# neighbour search (no gradient: indices)
# ------------------------------------------------------------------------------------------------

def knn_xyz(xyz, k, k2=0, drop_first=True):
    """(idx (B,N,k), idx2 (B,N,k2) or None): get_neighbor_index on COORDINATES, torch.topk's order among equal distances included
    (a tiled cloud, datasets/load_data.py:314-316, is full of them), for the two list lengths one resolution needs -- the layers'
    k and Pool_layer's k2 = 4 (gcn3d.py:236), which is not the prefix of the k-list on such a cloud.  Training and eval alike:
    a tie-free batch pays one small extra launch (csrc/knn_exact.hip)."""
    x = _req(xyz.detach(), torch.float32, "knn_xyz.xyz")
    B, N, C = x.shape
    if C != 3:
        raise HspError("knn_xyz: expects (B,N,3) coordinates")
    k2 = int(k2) if k2 and k2 < k else 0
    m = k + (1 if drop_first else 0)
    if m + 1 > 33 or N < 2 or N * 16 > 160 * 1024:         # beyond the tie pass's list length / its one-row-in-LDS bound
                                                            # (N > 10 240): the (distance, index) order, any N
        idx = knn(x, k, drop_first, _plain_xyz=True)
        return idx, (idx[:, :, :k2].contiguous() if k2 else None)
    idx = torch.empty(B, N, k, dtype=torch.int32, device=x.device)
    idx2 = torch.empty(B, N, k2, dtype=torch.int32, device=x.device) if k2 else None
    wsb = lib().hsp_knn_xyz_workspace_bytes(B, N)
    ws = _ws(wsb, x.device)
    _run("hsp_knn_xyz_f32", (_p(x), B, N, k, k2, 1 if drop_first else 0, _p(idx), _p(idx2), _p(ws), wsb, None, _stream()),
         key=f"B{B}N{N}k{k}", abytes=B * N * (12 + 4 * (k + k2)))
    return idx, idx2


def geometry_all(xyz, k0, kpool0, sel1, sel2, k1, kpool, k2):
    """``knn_xyz(xyz, k0, kpool0)`` AND ``geometry_levels`` in two launches instead of three (hsp_geometry_all_f32: the input cloud's
    tie pass rides in the levels' launch).  The dict of ``geometry_levels`` plus ``idx0`` / ``idx0_pool``; None outside the fused
    kernel's range."""
    x = _req(xyz.detach(), torch.float32, "geometry_all.xyz")
    s1, s2 = _req(sel1, torch.int32, "geometry_all.sel1"), _req(sel2, torch.int32, "geometry_all.sel2")
    B, N0, C = x.shape
    N1, N2 = s1.numel(), s2.numel()
    if (C != 3 or not (64 <= N2 <= N1 <= 576) or not (576 < N0 <= 1088) or B * N0 >= 131072 or k1 + 2 > 33 or k2 + 2 > 33 or k0 + 2 > 33
            or k1 + 1 > N1 or k2 + 1 > N2 or not 0 < kpool <= k1 or not 0 < kpool0 < k0):
        return None
    dev = x.device
    i32 = dict(dtype=torch.int32, device=dev)
    out = dict(v1=torch.empty(B, N1, 3, dtype=torch.float32, device=dev), v2=torch.empty(B, N2, 3, dtype=torch.float32, device=dev),
               idx0=torch.empty(B, N0, k0, **i32), idx0_pool=torch.empty(B, N0, kpool0, **i32),
               idx1=torch.empty(B, N1, k1, **i32), idx1_pool=torch.empty(B, N1, kpool, **i32), idx2=torch.empty(B, N2, k2, **i32),
               up1=torch.empty(B, N0, **i32), up2=torch.empty(B, N0, **i32))
    wsb = lib().hsp_geometry_all_workspace_bytes(B, N0)
    ws = _ws(wsb, dev)
    _run("hsp_geometry_all_f32", (_p(x), B, N0, k0, kpool0, _p(s1), N1, _p(s2), N2, k1, kpool, k2, 1, _p(out["idx0"]), _p(out["idx0_pool"]),
                                  _p(out["v1"]), _p(out["v2"]), _p(out["idx1"]), _p(out["idx1_pool"]), _p(out["idx2"]), _p(out["up1"]),
                                  _p(out["up2"]), _p(ws), wsb, _stream()),
         key=f"B{B}N{N0}/{N1}/{N2}k{k0}", abytes=B * (12 * N0 + 4 * (N0 * (k0 + kpool0) + N1 * (k1 + kpool + 3) + N2 * (k2 + 3)) + 8 * N0))
    return out


def geometry_levels(xyz, sel1, sel2, k1, kpool, k2):
    """the coordinate work of the two coarse levels in one launch (hsp_geometry_levels_f32): given the rows Pool_layer keeps at
    each level (sel1 of the input cloud, sel2 of level 1; int32 device vectors), returns a dict with the levels' vertices ``v1`` /
    ``v2``, the neighbour lists ``idx1`` (k1) / ``idx1_pool`` (kpool) / ``idx2`` (k2) and the nearest-point maps ``up1`` / ``up2``
    of the input cloud onto them (FaceRecon.py:100-101) -- or None when the shapes are outside the fused kernel's range (the
    caller then keeps the separate searches)."""
    x = _req(xyz.detach(), torch.float32, "geometry_levels.xyz")
    s1, s2 = _req(sel1, torch.int32, "geometry_levels.sel1"), _req(sel2, torch.int32, "geometry_levels.sel2")
    B, N0, C = x.shape
    N1, N2 = s1.numel(), s2.numel()
    if C != 3 or not (64 <= N2 <= N1 <= 576) or N1 > N0 or k1 + 2 > 33 or k2 + 2 > 33 or k1 + 1 > N1 or k2 + 1 > N2 or not 0 < kpool <= k1:
        return None
    dev = x.device
    out = dict(v1=torch.empty(B, N1, 3, dtype=torch.float32, device=dev), v2=torch.empty(B, N2, 3, dtype=torch.float32, device=dev),
               idx1=torch.empty(B, N1, k1, dtype=torch.int32, device=dev), idx1_pool=torch.empty(B, N1, kpool, dtype=torch.int32, device=dev),
               idx2=torch.empty(B, N2, k2, dtype=torch.int32, device=dev), up1=torch.empty(B, N0, dtype=torch.int32, device=dev),
               up2=torch.empty(B, N0, dtype=torch.int32, device=dev))
    _run("hsp_geometry_levels_f32", (_p(x), B, N0, _p(s1), N1, _p(s2), N2, k1, kpool, k2, 1, _p(out["v1"]), _p(out["v2"]), _p(out["idx1"]),
                                     _p(out["idx1_pool"]), _p(out["idx2"]), _p(out["up1"]), _p(out["up2"]), _stream()),
         key=f"B{B}N{N0}/{N1}/{N2}", abytes=B * (12 * N0 + 4 * (N1 * (k1 + kpool + 3) + N2 * (k2 + 3)) + 8 * N0))
    return out


def knn(x, k, drop_first=True, transposed_view=False, _plain_xyz=False):
    """int32 (B,N,k) nearest rows of x (B,N,C) per row; semantics of gcn3d.get_neighbor_index.  ``transposed_view`` (exact scope
    only): the reference holds these rows as the transposed view of a (B,C,N) tensor, which changes how its |x|^2 rounds."""
    x = _reqf(x.detach(), "knn.x")
    B, N, C = x.shape
    if C == 3 and x.dtype == torch.float32 and not _plain_xyz:
        return knn_xyz(x, k, 0, drop_first)[0]
    idx = torch.empty(B, N, k, dtype=torch.int32, device=x.device)
    L = lib()
    if x.dtype == torch.bfloat16:                           # feature rows stored in bf16: bf16 MFMA distance tiles
        if C == 3:
            raise HspError("knn: coordinates stay fp32 (gcn3d.py:57,59); bf16 is for feature rows")
        wsb = B * N * 4
        ws = _ws(wsb, x.device)
        _run("hsp_knn_bf16", (_p(x), B, N, C, k, 1 if drop_first else 0, _p(idx), _p(ws), wsb, _stream()),
             key=f"B{B}N{N}C{C}k{k}", abytes=B * N * (2 * C + 4 * k + 8), aflops=2 * B * N * N * C)
        return idx
    if _exact and k + (1 if drop_first else 0) + 1 <= 33 and os.environ.get("HSP_EXACT_TIES", "1") != "0":
        # eval-mode forward (exact_scope): torch.topk's own order among exactly equal distances (csrc/knn_exact.hip)
        wsb = L.hsp_knn_exact_workspace_bytes(B, N, C, k, 1 if drop_first else 0)
        ws = _ws(wsb, x.device)
        _run("hsp_knn_exact_f32", (_p(x), B, N, C, k, 1 if drop_first else 0, 1 if transposed_view else 0, _p(idx), _p(ws), wsb, None,
                                   _stream()),
             key=f"B{B}N{N}C{C}k{k}", abytes=B * N * (4 * C + 4 * k + (8 if C != 3 else 0)),
             aflops=(2 * B * N * N * C if C != 3 else 0))
        return idx
    wsb = L.hsp_knn_workspace_bytes(B, N, C, k)
    ws = _ws(wsb, x.device)
    _run("hsp_knn_f32", (_p(x), B, N, C, k, 1 if drop_first else 0, _p(idx), _p(ws), wsb, _stream()),
         key=f"B{B}N{N}C{C}k{k}", abytes=B * N * (4 * C + 4 * k + (8 if C != 3 else 0)),
         aflops=(2 * B * N * N * C if C != 3 else 0))       # feature path: the distance GEMM on the fp32 matrix cores
    return idx


_ext_state = None


def _ext_ok():
    """True when the compiled binding (hs_pose_amd/_hsp_torch.so) loads.  It only SHORTENS the host side of the no-grad inference
    forms -- one C++ call issues the launch sequence the ctypes route below issues from Python, the same libhsp.so kernels either
    way -- so a host where only libhsp.so builds (no torch / python headers for hsp_torch.cpp) still runs inference, through
    ctypes.  hs_pose_amd/chamfer.py, the reference extension's own surface, has no ctypes twin and raises."""
    global _ext_state
    if _ext_state is None:
        from . import _ext
        try:
            _ext_state = bool(_ext.available()) and _ext.ext() is not None
        except _ext.HspExtError:
            _ext_state = False
    return _ext_state


def center_cloud(points):
    """(points - mean over the points, mean (B,1,3)) with the mean in the reference's summation order (PoseNet9D.py:25)"""
    pts = _req(points.detach(), torch.float32, "center_cloud.points")
    if _timer is None and _ext_ok():
        from ._ext import ext
        return ext().center_cloud(pts)
    B, N, _ = pts.shape
    out = torch.empty_like(pts)
    mean = torch.empty(B, 1, 3, dtype=torch.float32, device=pts.device)
    _run("hsp_center_cloud_f32", (_p(pts), B, N, _p(out), _p(mean), _stream()), key=f"B{B}N{N}", abytes=24 * B * N)
    return out, mean


def nn1(target, source):
    """int32 (B,Nt) closest source row per target row; semantics of gcn3d.get_nearest_index."""
    t = _req(target.detach(), torch.float32, "nn1.target")
    s = _req(source.detach(), torch.float32, "nn1.source")
    B, Nt, C = t.shape
    if C != 3 or s.shape[2] != 3 or s.shape[0] != B:
        raise HspError("nn1: expects (B,Nt,3) and (B,Ns,3)")
    idx = torch.empty(B, Nt, dtype=torch.int32, device=t.device)
    _run("hsp_nn1_f32", (_p(t), Nt, _p(s), s.shape[1], B, _p(idx), _stream()),
         key=f"B{B}Nt{Nt}Ns{s.shape[1]}", abytes=B * (16 * Nt + 12 * s.shape[1]))
    return idx


def rev_index(idx, k, n_src):
    """reverse-edge (CSR) index (rev_off (B,n_src+1), rev_edge (B,Nq*k)) of the first k columns of the
    int32 neighbour index idx (B,Nq,kstride); memoised on the idx tensor, so a graph shared by several
    layers (the per-resolution xyz graph of the ORL branches) is inverted once per step."""
    cache = getattr(idx, "_hsp_rev", None)
    if cache is None:
        cache = {}
        idx._hsp_rev = cache
    hit = cache.get((k, n_src))
    if hit is not None:
        return hit
    if idx.dim() == 2:                                        # a row map (B,Nq): one edge per query
        (B, Nq), kstride = idx.shape, 1
    else:
        B, Nq, kstride = idx.shape
    off = torch.empty(B, n_src + 1, dtype=torch.int32, device=idx.device)
    edge = torch.empty(B, Nq * k, dtype=torch.int32, device=idx.device)
    _run("hsp_rev_build", (_p(idx), B, Nq, n_src, k, kstride, _p(off), _p(edge), _stream()),
         key=f"B{B}Nq{Nq}k{k}", abytes=B * (8 * Nq * k + 4 * n_src))
    cache[(k, n_src)] = (off, edge)
    return off, edge


# ------------------------------------------------------------------------------------------------
# receptive-field graph convolution
# ------------------------------------------------------------------------------------------------

class _RFSurface(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, idx, dirs_n, S):
        xyz = _req(xyz, torch.float32, "rf_surface.xyz")
        idx = _req(idx, torch.int32, "rf_surface.idx")
        dirs_n = _req(dirs_n, torch.float32, "rf_surface.dirs")
        B, N, k = idx.shape
        SC = dirs_n.shape[1]
        K = SC // S
        out = torch.empty(B, N, K, dtype=torch.float32, device=xyz.device)
        arg = torch.empty(B, N, SC, dtype=torch.uint16, device=xyz.device)
        _run("hsp_rf_surface_fwd", (_p(xyz), _p(idx), _p(dirs_n), B, N, k, S, K, _p(out), _p(arg), _stream()),
             key=f"B{B}N{N}k{k}S{S}C{K}", abytes=B * N * (12 + 4 * k + 4 * K + SC) + 12 * SC)
        ctx.save_for_backward(xyz, idx, dirs_n, arg)
        ctx.S = S
        return out

    @staticmethod
    def backward(ctx, g):
        xyz, idx, dirs_n, arg = ctx.saved_tensors
        g = _req(g, torch.float32, "rf_surface.grad")
        B, N, k = idx.shape
        SC = dirs_n.shape[1]
        gd = torch.empty_like(dirs_n)
        L = lib()
        wsb = L.hsp_rf_bwd_scatter_workspace_bytes(B, SC)
        ws = _ws(wsb, g.device)
        _rf_bwd_dirs_call("hsp_rf_surface_bwd", (_p(xyz), _p(dirs_n), _p(arg), _p(g), B, N, ctx.S, SC // ctx.S, _p(gd)), ws, wsb,
                          (dirs_n, gd), key=f"B{B}N{N}S{ctx.S}C{SC // ctx.S}", abytes=B * N * (12 + 4 * (SC // ctx.S) + SC) + 24 * SC)
        return None, None, gd, None


class _RFConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, idx, dirs_n, fm, S):
        xyz = _req(xyz, torch.float32, "rf_conv.xyz")
        idx = _req(idx, torch.int32, "rf_conv.idx")
        dirs_n = _req(dirs_n, torch.float32, "rf_conv.dirs")
        fm = _req(fm, torch.float32, "rf_conv.fm")
        B, N, k = idx.shape
        SC = dirs_n.shape[1]
        C = SC // S
        if fm.shape[-1] != (S + 1) * C:
            raise HspError("rf_conv: fm must have (S+1)*C columns")
        out = torch.empty(B, N, C, dtype=torch.float32, device=xyz.device)
        out, arg, fwin = _rf_conv_fwd_raw(xyz, idx, dirs_n, fm, S, any(ctx.needs_input_grad))
        ctx.save_for_backward(xyz, idx, dirs_n, fwin if fwin is not None else fm, arg)
        ctx.S = S
        return out

    @staticmethod
    def backward(ctx, g):
        xyz, idx, dirs_n, fm, arg = ctx.saved_tensors          # fm: the winners' support values (or fm itself
                                                                 # in DETERMINISTIC mode)
        g = _req(g, torch.float32, "rf_conv.grad")
        B, N, k = idx.shape
        SC = dirs_n.shape[1]
        C = SC // ctx.S
        gfm, gd = _rf_conv_bwd_raw(xyz, idx, dirs_n, fm, arg, g, ctx.S)
        return None, None, gd, gfm, None


def rf_surface(xyz, idx, directions, S):
    """mean_s max_n relu(R . normalize(directions, dim=0)): HSlayer_surface.graph_conv (gcn3d.py:92-107)
    -> (B,N,K).  ``directions`` is the RAW (3, S*K) parameter; its gradient includes the normalisation."""
    return _RFSurface.apply(xyz, idx, directions, S)


def rf_conv(xyz, idx, directions, fm, S):
    """HS_layer.graph_conv after the fm GEMM (gcn3d.py:166-181) -> (B,N,C); raw ``directions`` as above."""
    return _RFConv.apply(xyz, idx, directions, fm, S)


# ------------------------------------------------------------------------------------------------
# neighbourhood max-pool
# ------------------------------------------------------------------------------------------------

class _GatherMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, idx, qsel, k):
        feat = _reqf(feat, "gather_max.feat")
        idx = _req(idx, torch.int32, "gather_max.idx")
        if qsel is not None:
            qsel = _req(qsel, torch.int32, "gather_max.qsel")
        B, Nsrc, C = feat.shape
        Nidx, kstride = idx.shape[1], idx.shape[2]
        Nq = qsel.numel() if qsel is not None else Nidx
        out = torch.empty(B, Nq, C, dtype=feat.dtype, device=feat.device)
        arg = torch.empty(B, Nq, C, dtype=torch.uint8, device=feat.device)
        es = _es(feat)
        _run("hsp_gather_max_fwd" + _sfx(feat), (_p(feat), _p(idx), _p(qsel), B, Nsrc, Nidx, Nq, k, kstride, C, _p(out),
                                                 _p(arg), _stream()),
             key=f"B{B}Ns{Nsrc}Nq{Nq}k{k}C{C}", abytes=B * (es * Nsrc * C + Nq * (4 * k + (es + 1) * C)))
        ctx.save_for_backward(idx, qsel, arg)
        ctx.dims = (B, Nsrc, Nidx, Nq, kstride, C)
        return out

    @staticmethod
    def backward(ctx, g):
        idx, qsel, arg = ctx.saved_tensors
        B, Nsrc, Nidx, Nq, kstride, C = ctx.dims
        g = _reqf(g, "gather_max.grad")
        gfeat = torch.empty(B, Nsrc, C, dtype=g.dtype, device=g.device)
        es = _es(g)
        _run("hsp_gather_max_bwd" + _sfx(g), (_p(g), 0, _p(idx), _p(qsel), _p(arg), B, Nsrc, Nidx, Nq, kstride, C, _p(gfeat),
                                              0, _vp(0), _stream()),
             key=f"B{B}Ns{Nsrc}Nq{Nq}C{C}", abytes=B * (es * Nsrc * C + Nq * (4 * kstride + (es + 1) * C)))
        return gfeat, None, None, None


class _OrlGlobal(torch.autograd.Function):
    """fg[b,c] = mean_i max_n feat[b, idx[b,i,n], c]  (get_ORL_global, gcn3d.py:211-218, before the repeat)."""

    @staticmethod
    def forward(ctx, feat, idx, k):
        feat = _req(feat, torch.float32, "orl.feat")
        idx = _req(idx, torch.int32, "orl.idx")
        B, N, C = feat.shape
        kstride = idx.shape[2]
        fg, arg = _orl_fwd_raw(feat, idx, k)
        ctx.save_for_backward(idx, arg)
        ctx.dims = (B, N, kstride, C)
        ctx.k = k
        return fg

    @staticmethod
    def backward(ctx, gfg):
        idx, arg = ctx.saved_tensors
        B, N, kstride, C = ctx.dims
        g = _req(gfg / N, torch.float32, "orl.grad")
        gfeat = torch.empty(B, N, C, dtype=torch.float32, device=g.device)
        if DETERMINISTIC:      # gather form over the reverse-edge index (shared by the layers of a resolution)
            off, edge = rev_index(idx, ctx.k, N)
            _run("hsp_gather_max_bwd_csr", (_p(g), 1, _p(arg), _p(off), _p(edge), B, N, N, ctx.k, C, _p(gfeat), _stream()),
                 key=f"B{B}Ns{N}Nq{N}k{ctx.k}C{C}bc", abytes=B * N * (4 * C + 8 * ctx.k + C))
        else:                  # LDS tile kernel; broadcast gradient => integer counts => reproducible as well
            _run("hsp_gather_max_bwd", (_p(g), 1, _p(idx), _vp(0), _p(arg), B, N, N, N, kstride, C, _p(gfeat), 0, _vp(0),
                                        _stream()),
                 key=f"B{B}Ns{N}Nq{N}C{C}bc", abytes=B * N * (4 * C + 4 * kstride + C))
        return gfeat, None, None


class _PointsMax(torch.autograd.Function):
    """(B,N,C) -> (B,C) max over the points of each cloud; the gradient goes to the first winning row."""

    @staticmethod
    def forward(ctx, x):
        x = _req(x, torch.float32, "points_max.x")
        B, N, C = x.shape
        out = torch.empty(B, C, dtype=torch.float32, device=x.device)
        arg = torch.empty(B, C, dtype=torch.int32, device=x.device)
        _run("hsp_points_max_fwd", (_p(x), B, N, C, _p(out), _p(arg), _stream()), key=f"B{B}N{N}C{C}",
             abytes=B * (4 * N * C + 8 * C))
        ctx.save_for_backward(arg)
        ctx.dims = (B, N, C)
        return out

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        B, N, C = ctx.dims
        g = _req(g, torch.float32, "points_max.grad")
        gx = torch.empty(B, N, C, dtype=torch.float32, device=g.device)
        _run("hsp_points_max_bwd", (_p(g), _p(arg), B, N, C, _p(gx), _stream()), key=f"B{B}N{N}C{C}",
             abytes=B * (4 * N * C + 8 * C))
        return gx


class _PoolLayer(torch.autograd.Function):
    """Pool_layer.forward (gcn3d.py:236-245) in one launch: (max over the k nearest of the kept rows, the kept rows' xyz)."""

    @staticmethod
    def forward(ctx, feat, xyz, idx, qsel, k):
        feat = _req(feat, torch.float32, "pool.feat")
        xyz = _req(xyz, torch.float32, "pool.xyz")
        idx = _req(idx, torch.int32, "pool.idx")
        qsel = _req(qsel, torch.int32, "pool.qsel")
        B, N, C = feat.shape
        Nq, kstride = qsel.numel(), idx.shape[2]
        out = torch.empty(B, Nq, C, dtype=torch.float32, device=feat.device)
        arg = torch.empty(B, Nq, C, dtype=torch.uint8, device=feat.device)
        vsel = torch.empty(B, Nq, 3, dtype=torch.float32, device=feat.device)
        _run("hsp_pool_fwd", (_p(feat), _p(xyz), _p(idx), _p(qsel), B, N, Nq, k, kstride, C, _p(out), _p(arg), _p(vsel), _stream()),
             key=f"B{B}N{N}Nq{Nq}k{k}C{C}", abytes=B * (4 * N * C + Nq * (4 * k + 5 * C + 24)))
        ctx.save_for_backward(idx, qsel, arg)
        ctx.dims = (B, N, idx.shape[1], Nq, kstride, C)
        ctx.mark_non_differentiable(vsel)
        ctx.set_materialize_grads(False)          # (else autograd fills a zero gradient for vsel every step: two fill kernels)
        return out, vsel

    @staticmethod
    def backward(ctx, g, _gv):
        idx, qsel, arg = ctx.saved_tensors
        B, Nsrc, Nidx, Nq, kstride, C = ctx.dims
        g = _req(g, torch.float32, "pool.grad")
        gfeat = torch.empty(B, Nsrc, C, dtype=torch.float32, device=g.device)
        _run("hsp_gather_max_bwd", (_p(g), 0, _p(idx), _p(qsel), _p(arg), B, Nsrc, Nidx, Nq, kstride, C, _p(gfeat), 0, _vp(0),
                                    _stream()),
             key=f"B{B}Ns{Nsrc}Nq{Nq}C{C}", abytes=B * (4 * Nsrc * C + Nq * (4 * kstride + 5 * C)))
        return gfeat, None, None, None, None


def pool_layer(feat, xyz, idx, qsel, k):
    """(feature_map_pool (B,Nq,C), vertices_pool (B,Nq,3)) of Pool_layer for fp32 rows; xyz carries no gradient."""
    if (_timer is None and not torch.is_grad_enabled() and feat.is_cuda and feat.dtype == torch.float32 and feat.is_contiguous()
            and xyz.dtype == torch.float32 and xyz.is_contiguous() and idx.dtype == torch.int32 and idx.is_contiguous()
            and qsel.dtype == torch.int32 and qsel.dim() == 1 and qsel.is_contiguous() and _ext_ok()):
        from ._ext import ext
        v, out = ext().pool_forward(xyz, feat, idx, qsel, k)
        return out, v
    return _PoolLayer.apply(feat, xyz, idx, qsel, k)


def points_max(x):
    """max over dim 1 of a point-major (B,N,C) tensor -> (B,C)   (the heads' torch.max(x, 2)[0])."""
    return _PointsMax.apply(x)


def gather_max(feat, idx, k, qsel=None):
    """max over the first k listed neighbours of each (selected) row -> (B,Nq,C)."""
    return _GatherMax.apply(feat, idx, qsel, k)


def orl_global(feat, idx, k):
    """(B,C) outlier-robust global feature (mean over points of the neighbourhood max)."""
    return _OrlGlobal.apply(feat, idx, k)


# ------------------------------------------------------------------------------------------------
# whole HS layers as ONE autograd node each (hand-written backward)
#
# Autograd over the small ops above costs ~25 tiny add / copy / reduce kernels per layer and step; here
# the layer is laid out as its minimal kernel sequence and every gradient sum is an in-place GEMM
# accumulation (addmm_) or a kernel epilogue:
#   forward   fm = X W + b ; F = graph_conv(fm) ; fg = mean_i max_n F[idx_xyz] ; t = fg Wb^T
#             out = F + F Wa^T + X Wste^T + t[b]                  (conv2 = [Wa | Wb], gcn3d.py:109-113,183-187)
#   backward  gt = sum_i g ; gWb = gt^T fg ; gfg = gt Wb
#             gF = g + g Wa  (+= ORL scatter of gfg/N, accumulated by the kernel) ; gWa = g^T F
#             gfm, gD = graph_conv_bwd(gF) ; gW = X^T gfm ; gb = colsum(gfm)
#             gX = g Wste + gfm W^T ; gWste = g^T X
# ------------------------------------------------------------------------------------------------

def _wgrad_fallback(A2, B2, out, colsum):
    """A2^T B2 for the shapes the split-K kernels do not take (an output dimension that is no multiple of 64 and too small for
    the ragged form, e.g. a 3-wide coordinate block): the general tile kernel on a transposed copy of A2"""
    res = gemm_rows(A2.t().contiguous(), B2 if B2.stride(1) == 1 else B2.contiguous(), nn1=True)
    out.copy_(res)
    if not colsum:
        return out
    if B2.is_contiguous():                                   # two-stage kernel: no ATen multi-block reduction
        return out, colsum_rows(B2.view(1, B2.shape[0], B2.shape[1])).view(-1)
    return out, B2.sum(dim=0)


class WgradBatch:
    """Folds of several weight gradients in one launch: inside ``with WgradBatch():`` the custom ``wgrad`` runs only its split-K
    launch and leaves the fold pending (the outputs are NOT valid until the block exits); the exit folds up to four pending
    problems per ``hsp_wgrad_fold`` launch.  An HS layer's backward computes three parameter gradients: two kernel boundaries
    fewer per layer, same fixed summation order, same bits."""
    current = None

    def __init__(self):
        self.items = []                   # (HspWgradPending, workspace tensor kept alive)

    def __enter__(self):
        self.prev, WgradBatch.current = WgradBatch.current, self
        return self

    def __exit__(self, *exc):
        WgradBatch.current = self.prev
        if exc[0] is None:
            if StepFolds.current is not None:          # the whole backward's folds go out together (StepFolds)
                StepFolds.current.wgrads.extend(self.items)
                self.items = []
            else:
                self.flush()
        return False

    def flush(self):
        from ._lib import HspWgradPending
        cap = StepFolds.MAX_WGRAD
        while self.items:
            chunk, self.items = self.items[:cap], self.items[cap:]
            arr = (HspWgradPending * len(chunk))(*[c[0] for c in chunk])
            _run("hsp_wgrad_fold", (arr, len(chunk), _stream()), key=f"n{len(chunk)}")


class StepFolds:
    """Every fold a backward pass leaves pending, in ONE launch at its end (``hsp_step_fold``): the split-K folds of the
    parameter gradients (``WgradBatch`` hands its items over instead of flushing per layer) and the per-cloud folds of the
    receptive-field layers' direction gradients.  Nothing before the optimizer reads those results, and each fold is a 4-8 us
    dependent launch: an HS stack's backward has ten.  Same summation order as the stand-alone folds: same bits
    (tests/test_gpu_step_folds.py).

    ``with StepFolds(): loss.backward()`` -- the parameter gradients are NOT valid until the block exits, so the scope is only
    for callers that own the whole backward and start it with ``p.grad = None`` for every parameter (autograd then keeps the
    very tensors the nodes return; an accumulation into an existing ``.grad`` would read them before the fold):
    ``graph.GraphedStep`` and ``graph.GraphedNetwork``, whose captures replay the fold with the rest of the step."""
    current = None
    MAX_WGRAD, MAX_DIRS = 24, 8        # include/hsp.h: HSP_FOLD_MAX_WGRAD, HSP_FOLD_MAX_DIRS

    def __init__(self, bare_wgrad=False):
        """bare_wgrad: also defer ``wgrad`` calls made outside a ``WgradBatch`` (``linear_rows``' backward, which returns a
        TRANSPOSED view of its result: fine under ``torch.autograd.grad``, which hands the view back, not under
        ``.backward()``, whose gradient accumulator clones a view the moment it arrives)."""
        self.bare_wgrad = bare_wgrad
        self.wgrads = []                  # (HspWgradPending, workspace kept alive)
        self.dirs = []                    # (HspDirsPending, workspace, directions, grad kept alive)

    def __enter__(self):
        if StepFolds.current is not None:
            raise HspError("StepFolds: scopes do not nest")
        StepFolds.current = self
        return self

    def __exit__(self, *exc):
        StepFolds.current = None
        if exc[0] is None:
            self.flush()
        else:
            self.wgrads, self.dirs = [], []
        return False

    def flush(self):
        from ._lib import HspWgradPending, HspDirsPending
        while self.wgrads or self.dirs:
            w, self.wgrads = self.wgrads[:self.MAX_WGRAD], self.wgrads[self.MAX_WGRAD:]
            d, self.dirs = self.dirs[:self.MAX_DIRS], self.dirs[self.MAX_DIRS:]
            wa = (HspWgradPending * max(len(w), 1))(*[c[0] for c in w])
            da = (HspDirsPending * max(len(d), 1))(*[c[0] for c in d])
            _run("hsp_step_fold", (wa, len(w), da, len(d), _stream()), key=f"w{len(w)}d{len(d)}")


def _hold(t):
    """keep a pending fold's output MEMORY alive without holding the tensor: autograd's gradient accumulator keeps the very
    tensor a node returns only while nobody else references it (it clones otherwise, and the clone would be taken before
    the fold has run); the storage object pins the bytes, not the tensor."""
    return None if t is None else t.untyped_storage()


def _rf_bwd_dirs_call(name, args_before_ws, ws, wsb, keep, key, abytes):
    """run a receptive-field backward entry point ``name`` (args ..., ws, ws_bytes, stream); inside a ``StepFolds`` scope its
    ``_partial`` form, whose direction-gradient fold goes out with the step's other folds.  keep: tensors the pending fold
    reads / writes (kept alive until it has run)."""
    sf = StepFolds.current
    if sf is None:
        _run(name, (*args_before_ws, _p(ws), wsb, _stream()), key=key, abytes=abytes)
        return
    from ._lib import HspDirsPending
    pend = HspDirsPending()
    pname = name.replace("_bf16", "") + "_partial" + ("_bf16" if name.endswith("_bf16") else "")
    _run(pname, (*args_before_ws, _p(ws), wsb, ctypes.byref(pend), _stream()), key=key, abytes=abytes)
    sf.dirs.append((pend, ws) + tuple(_hold(t) for t in keep))


def _wgrad_custom(A2, B2, out, colsum):
    K, M = A2.shape
    N = B2.shape[1]
    cs = torch.empty(N, dtype=torch.float32, device=A2.device) if colsum else None
    L = lib()
    wsb = L.hsp_wgrad_workspace_bytes(M, N, K)
    ws = _ws(wsb, A2.device)
    sfx = "bf16" if A2.dtype == torch.bfloat16 else "f32"
    es = 2 if A2.dtype == torch.bfloat16 else 4
    batch = WgradBatch.current
    sf = StepFolds.current
    sink = batch.items if batch is not None else sf.wgrads if sf is not None and sf.bare_wgrad else None
    if sink is not None:
        from ._lib import HspWgradPending
        pend = HspWgradPending()
        _run("hsp_wgrad_partial_" + sfx, (_p(A2), A2.stride(0), _p(B2), B2.stride(0), M, N, K, _p(out), out.stride(0), _p(cs),
                                          _p(ws), wsb, ctypes.byref(pend), _stream()),
             key=f"M{M}N{N}K{K}", abytes=es * K * (M + N) + 4 * M * N, aflops=2 * M * N * K)
        sink.append((pend, ws, _hold(out), _hold(cs)))
    else:
        _run("hsp_wgrad_" + sfx, (_p(A2), A2.stride(0), _p(B2), B2.stride(0), M, N, K, _p(out), out.stride(0), _p(cs),
                                  _p(ws), wsb, _stream()),
             key=f"M{M}N{N}K{K}", abytes=es * K * (M + N) + 4 * M * N, aflops=2 * M * N * K)
    return (out, cs) if colsum else out


def wgrad(A2, B2, out=None, colsum=False):
    """A2^T @ B2 for point-row matrices A2 (K,M), B2 (K,N) (rows may be strided views of wider tensors)
    -> (M,N) [+ column sums of B2 = the bias gradient]: the parameter-gradient GEMM on the split-K kernels of csrc/gemm.hip
    (column sum fused, strided output, bit-reproducible); shapes they do not cover go to the general tile kernel."""
    K, M = A2.shape
    N = B2.shape[1]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A2.device)
    if _wgrad_ragged_ok(A2, B2, out) or _wgrad_ok(A2, B2, out):
        return _wgrad_custom(A2, B2, out, colsum)
    return _wgrad_fallback(A2, B2, out, colsum)


def _wgrad_ragged_ok(A2, B2, out):
    """A2^T B2 with M = A2.shape[1] NOT a multiple of 64 (the heads' first layers: K = 1286 / 1289 / 771 input columns,
    PoseR.py:27, PoseTs.py:32, FaceRecon.py:38,116) on the x3 kernel: fp32 rows of A2 on a 16-byte pitch that covers ceil4(M)
    (``assemble_feat`` / ``cat_rows_pitched`` lay them out so), N a multiple of 128"""
    K, M = A2.shape
    N = B2.shape[1]
    return (M % 64 != 0 and M >= 128 and N % 128 == 0
            and -(-M // 128) * (N // 128) >= 4 and A2.dtype == torch.float32 and B2.dtype == torch.float32 and A2.is_cuda
            and A2.stride(1) == 1 and B2.stride(1) == 1 and out.stride(1) == 1 and A2.stride(0) % 4 == 0 and B2.stride(0) % 4 == 0
            and A2.stride(0) >= (M + 3) // 4 * 4 and A2.data_ptr() % 16 == 0 and B2.data_ptr() % 16 == 0)


def _wgrad_ok(A2, B2, out):
    K, M = A2.shape
    N = B2.shape[1]
    return (M % 64 == 0 and N % 64 == 0 and A2.stride(1) == 1 and B2.stride(1) == 1 and out.stride(1) == 1
            and A2.stride(0) % 2 == 0 and B2.stride(0) % 2 == 0 and A2.dtype == torch.float32 and A2.is_cuda)


def wgrad_pair(A0, B0, out0, A1, B1, out1):
    """(A0^T B0 -> out0, A1^T B1 -> out1) inside a ``WgradBatch``: ONE split-K launch for both when each is a K-sliced problem the
    hand-written kernel takes (the two parameter gradients of an HS layer that depend only on the incoming gradient: g^T F and
    g^T X -- each alone leaves most of the chip idle); otherwise two ``wgrad`` calls."""
    batch = WgradBatch.current
    if batch is None or not _wgrad_ok(A0, B0, out0) or not _wgrad_ok(A1, B1, out1):
        wgrad(A0, B0, out=out0)
        wgrad(A1, B1, out=out1)
        return
    from ._lib import HspWgradPending
    L = lib()
    (K0, M0), N0 = A0.shape, B0.shape[1]
    (K1, M1), N1 = A1.shape, B1.shape[1]
    wsb0, wsb1 = L.hsp_wgrad_workspace_bytes(M0, N0, K0), L.hsp_wgrad_workspace_bytes(M1, N1, K1)
    ws0, ws1 = _ws(wsb0, A0.device), _ws(wsb1, A0.device)
    pend = (HspWgradPending * 2)()
    _run("hsp_wgrad_partial_pair_f32", (_p(A0), A0.stride(0), _p(B0), B0.stride(0), M0, N0, K0, _p(out0), out0.stride(0), _p(ws0), wsb0,
                                        _p(A1), A1.stride(0), _p(B1), B1.stride(0), M1, N1, K1, _p(out1), out1.stride(0), _p(ws1), wsb1,
                                        pend, _stream()),
         key=f"M{M0}N{N0}K{K0}+M{M1}N{N1}K{K1}", abytes=4 * (K0 * (M0 + N0) + M0 * N0 + K1 * (M1 + N1) + M1 * N1),
         aflops=2 * (M0 * N0 * K0 + M1 * N1 * K1))
    batch.items.append((HspWgradPending.from_buffer_copy(pend[0]), ws0, _hold(out0)))
    batch.items.append((HspWgradPending.from_buffer_copy(pend[1]), ws1, _hold(out1)))


def _ld(t):
    """leading dimension (elements) of a 2-D tensor whose rows are contiguous"""
    if t.dim() != 2 or t.stride(1) != 1:
        raise HspError("gemm_rows: operands must be 2-D with contiguous rows (stride(1) == 1)")
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def gemm_rows(A1, B1, nn1=False, A2=None, B2=None, nn2=False, bias=None, resid=None, cloud_bias=None, rows_per_cloud=0,
              out=None, alpha=1.0, xyz3=None, w3=None):
    """out (M,N) = alpha * (A1 @ op(B1) [+ A2 @ op(B2)]) [+ bias] [+ resid] [+ cloud_bias[row // rows_per_cloud]]   (csrc/gemm_rows.hip)

    op(B) = B^T for a (N,K) weight (``nn=False``: a Linear / Conv1d(k=1) weight) or B for a (K,N) matrix (``nn=True``:
    HS_layer.weights).  fp32 or bf16 (all of A, B, resid, out alike; bias / cloud_bias always fp32; bf16 takes (N,K)
    operands only).  Operands may be row-strided views (column blocks of wider tensors).  ``xyz3`` (M,3) / ``w3`` (N,3)
    fp32: adds xyz3[row] . w3[col] -- the K = 3 STE of HSlayer_surface on raw coordinates."""
    dt = A1.dtype
    if dt not in (torch.float32, torch.bfloat16):
        raise HspError(f"gemm_rows: fp32 or bf16 operands, got {dt}")
    for t_, nm in ((A1, "A1"), (B1, "B1"), (A2, "A2"), (B2, "B2"), (resid, "resid")):
        if t_ is not None and (not t_.is_cuda or t_.dtype != dt):
            raise HspError(f"gemm_rows.{nm}: expected a {dt} GPU tensor")
    if out is not None and (not out.is_cuda or out.dtype not in (dt, torch.float32)):
        raise HspError(f"gemm_rows.out: expected a {dt} (or, for bf16 operands, fp32) GPU tensor")
    M, K1 = A1.shape
    N = B1.shape[1] if nn1 else B1.shape[0]
    if (B1.shape[0] if nn1 else B1.shape[1]) != K1:
        raise HspError("gemm_rows: A1 / B1 inner dimensions differ")
    K2 = 0
    if A2 is not None:
        K2 = A2.shape[1]
        if A2.shape[0] != M or (B2.shape[1] if nn2 else B2.shape[0]) != N or (B2.shape[0] if nn2 else B2.shape[1]) != K2:
            raise HspError("gemm_rows: second source has the wrong shape")
    if out is None:
        out = torch.empty(M, N, dtype=dt, device=A1.device)
    for t_ in (bias, cloud_bias, xyz3, w3):
        if t_ is not None and (t_.dtype != torch.float32 or not t_.is_cuda or not t_.is_contiguous()):
            raise HspError("gemm_rows: bias / cloud_bias / xyz3 / w3 must be contiguous fp32 GPU tensors")
    if (xyz3 is None) != (w3 is None) or (xyz3 is not None and (xyz3.shape != (M, 3) or w3.shape != (N, 3))):
        raise HspError("gemm_rows: xyz3 (M,3) and w3 (N,3) go together")
    flops = 2 * M * N * (K1 + K2)
    es = 4 if dt == torch.float32 else 2
    ab = es * (M * (K1 + K2) + N * (K1 + K2) + M * N * (2 if resid is not None else 1))
    key = f"M{M}N{N}K{K1}" + (f"+{K2}" if K2 else "")
    wsb = lib().hsp_gemm_rows_workspace_bytes(M, N, K1, K2, es)       # split-K partial tiles (0 for most shapes)
    ws = _ws(wsb, A1.device) if wsb else None
    if dt == torch.float32:
        _run("hsp_gemm_rows_f32", (_p(A1), _ld(A1), _p(B1), _ld(B1), 1 if nn1 else 0, K1,
                                   _p(A2), _ld(A2) if A2 is not None else 0, _p(B2), _ld(B2) if B2 is not None else 0,
                                   1 if nn2 else 0, K2, M, N, _p(bias), _p(resid), _ld(resid) if resid is not None else 0,
                                   _p(cloud_bias), int(rows_per_cloud), float(alpha), _p(xyz3), _p(w3), _p(out), _ld(out),
                                   _p(ws), wsb, _stream()),
             key=key, abytes=ab, aflops=flops)
    else:
        if nn1 or nn2:
            raise HspError("gemm_rows: bf16 operands must be (N,K) (pass the transposed copy)")
        _run("hsp_gemm_rows_bf16", (_p(A1), _ld(A1), _p(B1), _ld(B1), K1,
                                    _p(A2), _ld(A2) if A2 is not None else 0, _p(B2), _ld(B2) if B2 is not None else 0, K2,
                                    M, N, _p(bias), _p(resid), _ld(resid) if resid is not None else 0, _p(cloud_bias),
                                    int(rows_per_cloud), float(alpha), _p(xyz3), _p(w3), _p(out), _ld(out),
                                    1 if out.dtype == torch.float32 else 0, _p(ws), wsb, _stream()),
             key=key + "bf16", abytes=ab, aflops=flops)
    return out


def gemm_wave_supported(M, N, K1, K2=0, cfg=0):
    return bool(lib().hsp_gemm_wave_supported(int(M), int(N), int(K1), int(K2), int(cfg)))


def gemm_wave(A1, B1, nn1=False, A2=None, B2=None, nn2=False, bias=None, resid=None, cloud_bias=None, rows_per_cloud=0,
              out=None, alpha=1.0, xyz3=None, w3=None, cfg=0):
    """the contract of ``gemm_rows`` (fp32 only) on the LDS-free wave-level kernel (csrc/gemm_wave.hip): K1, K2 multiples of
    32, N of 32, 16-byte aligned rows.  ``cfg`` != 0 forces a tile / cut (include/hsp.h), for tuning."""
    M, K1 = A1.shape
    N = B1.shape[1] if nn1 else B1.shape[0]
    K2 = A2.shape[1] if A2 is not None else 0
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A1.device)
    flops = 2 * M * N * (K1 + K2)
    ab = 4 * (M * (K1 + K2) + N * (K1 + K2) + M * N * (2 if resid is not None else 1))
    _run("hsp_gemm_wave_f32", (_p(A1), _ld(A1), _p(B1), _ld(B1), 1 if nn1 else 0, K1,
                               _p(A2), _ld(A2) if A2 is not None else 0, _p(B2), _ld(B2) if B2 is not None else 0,
                               1 if nn2 else 0, K2, M, N, _p(bias), _p(resid), _ld(resid) if resid is not None else 0,
                               _p(cloud_bias), int(rows_per_cloud), float(alpha), _p(xyz3), _p(w3), _p(out), _ld(out),
                               int(cfg), _stream()),
         key=f"M{M}N{N}K{K1}" + (f"+{K2}" if K2 else ""), abytes=ab, aflops=flops)
    return out


# ------------------------------------------------------------------------------------------------
# fp32 products on the bf16 matrix cores (csrc/gemm_x3.hip): the weight-side operand is kept as three exact bf16 slices in
# (N, K) form.  ``X3Planes`` owns the slices of every weight matrix the products have asked for; ``refresh()`` re-splits all
# of them from the fp32 masters in ONE launch (FaceRecon / PoseNet9D call it at the start of a forward: the optimizer has
# moved the weights since the last one), a matrix seen for the first time is registered and split on the spot.
# ------------------------------------------------------------------------------------------------
class X3Planes:
    """The registry of ONE network (FaceRecon / PoseNet9D own one and make it current for their forward; the autograd nodes
    remember it for their backward).  Entries keep a reference to the weight view they split, so an address cannot be reused by
    another matrix while its entry lives; the registry -- tables, planes and references -- goes away with the network.  A table
    that was replaced (a matrix registered later) is parked, not freed, ONCE a hipGraph capture has read this registry: the captured
    graph may still point to it.  A registry no capture has touched (ad-hoc callers, gradient checks) frees what it replaces, so the
    ``max_entries`` bound of the process-wide registry does bound device memory."""

    def __init__(self, max_entries=0):
        self.entries = {}            # key -> dict(src (kept alive), planes, N, K, kp, transpose, ver: src._version at the split)
        self.table = None            # device table of every entry (rebuilt when an entry is added)
        self.total_tiles = 0
        self._old_tables = []
        self._captured = False       # a capture has read planes / the table of this registry
        self.max_entries = max_entries

    def _park(self, t):
        if self._captured:
            self._old_tables.append(t)

    @staticmethod
    def _key(W, transpose):
        return (W.data_ptr(), tuple(W.shape), W.stride(0), bool(transpose))

    def planes(self, W, transpose):
        """(planes (3, N, kp) bf16, kp, plane stride) of the fp32 matrix W: (N, K) rows, or (K, N) rows with ``transpose``"""
        k_ = self._key(W, transpose)
        e = self.entries.get(k_)
        if not self._captured and torch.cuda.is_current_stream_capturing():
            self._captured = True
        if e is None:
            if torch.cuda.is_current_stream_capturing():
                raise HspError("gemm_x3: a weight matrix was first seen inside a graph capture (run one eager step first)")
            if W.dim() != 2 or W.stride(1) != 1 or W.dtype != torch.float32 or not W.is_cuda:
                raise HspError("gemm_x3: weights must be 2-D fp32 GPU matrices with contiguous rows")
            N, K = (W.shape[1], W.shape[0]) if transpose else (W.shape[0], W.shape[1])
            kp = (K + 31) // 32 * 32
            e = dict(src=W.detach(), planes=torch.zeros(3, N, kp, dtype=torch.bfloat16, device=W.device), N=N, K=K, kp=kp,
                     transpose=bool(transpose), ver=W._version)
            if self.max_entries and len(self.entries) >= self.max_entries:      # the process-wide registry of ad-hoc callers:
                # oldest out (a network's own registry is unbounded) -- its planes are parked like a replaced table, not freed:
                # a captured hipGraph may still read them
                self._park(self.entries.pop(next(iter(self.entries)))["planes"])
            self.entries[k_] = e
            self._split([e])                                   # this matrix now ...
            if self.table is not None:
                self._park(self.table)
            self.table, self.total_tiles = self._table_of(list(self.entries.values()))   # ... and the table of all for refresh()
        elif e["ver"] != e["src"]._version and not torch.cuda.is_current_stream_capturing():
            # the weights moved in place (optimizer.step, load_state_dict) since these planes were cut and nobody called
            # refresh(): a stand-alone module / a direct ops.linear_rows caller.  Re-split this matrix now.
            self._split([e])
            e["ver"] = e["src"]._version
        return e["planes"], e["kp"], e["N"] * e["kp"]

    def _table_of(self, ents):
        from ._lib import HspSplitDesc
        tab = (HspSplitDesc * len(ents))()
        tile0 = 0
        for i, e in enumerate(ents):
            W = e["src"]
            tab[i].src, tab[i].dst = W.data_ptr(), e["planes"].data_ptr()
            tab[i].rows, tab[i].cols, tab[i].ld = W.shape[0], W.shape[1], W.stride(0)
            tab[i].transpose, tab[i].kp, tab[i].tile0, tab[i].ps = int(e["transpose"]), e["kp"], tile0, e["N"] * e["kp"]
            tile0 += ((e["N"] + 31) // 32) * ((e["K"] + 63) // 64)        # pieces of 32 (n) x 64 (k) output elements
        host = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8)
        return host.to(ents[0]["src"].device), tile0

    def _split(self, ents):
        tab, tiles = self._table_of(ents)
        _run("hsp_split_params_x3", (_p(tab), len(ents), tiles, _stream()), key=f"n{len(ents)}")
        return tab

    def refresh(self):
        if not self.entries:
            return
        if not self._captured and torch.cuda.is_current_stream_capturing():
            self._captured = True
        ents = list(self.entries.values())
        _run("hsp_split_params_x3", (_p(self.table), len(ents), self.total_tiles, _stream()), key=f"n{len(ents)}",
             abytes=sum(10 * e["N"] * e["K"] for e in ents))
        for e in ents:
            e["ver"] = e["src"]._version


x3_planes = X3Planes(max_entries=256)   # the CURRENT registry (a process-wide one until a network installs its own: x3_scope)


class x3_scope:
    """``with ops.x3_scope(reg):`` makes ``reg`` the registry the x3 products look their weight planes up in"""

    def __init__(self, reg):
        self.reg = reg

    def __enter__(self):
        global x3_planes
        self.prev, x3_planes = x3_planes, (self.reg if self.reg is not None else x3_planes)
        return self.reg

    def __exit__(self, *exc):
        global x3_planes
        x3_planes = self.prev
        return False


GEMM_X3 = True         # False (tools/gemm_gap.py, profiling): the products stay on the fp32 matrix cores (gemm_wave / gemm_rows)


def x3_refresh():
    """re-split every weight matrix of the current registry (one launch); call once per forward, before the first product"""
    if GEMM_X3:
        x3_planes.refresh()


def gemm_x3_ok(A1, B1, A2, B2, bias, resid, cloud_bias, xyz3, out, M, N):
    if not GEMM_X3 or xyz3 is not None or N < 64 or A1.dtype != torch.float32:
        return False
    K2 = A2.shape[1] if A2 is not None else 0
    if not lib().hsp_gemm_x3_supported(M, N, A1.shape[1], K2):
        return False
    epi = (1 if bias is not None else 0) | (2 if resid is not None else 0) | (4 if cloud_bias is not None else 0)
    if epi not in (0, 1, 6) and not (epi in (2, 4) and A2 is None):
        return False
    if epi and ((M + 63) // 64) * ((N + 127) // 128) < 128:            # (an epilogue rules out split-K: too few workgroups)
        return False
    for t_ in (A1, A2):
        if t_ is not None and (t_.data_ptr() % 16 or (t_.stride(0) * 4) % 16):
            return False
    return True


def gemm_x3_bn(A1, B1, A2, B2, resid, cloud_bias, rows_per_cloud, out):
    """the layer's out product  A1 B1^T + A2 B2^T + resid + cloud_bias[cloud]  on csrc/gemm_x3.hip that also leaves the first pass
    of the BatchNorm that follows: returns ONE tensor (1 + 2 tiles, N): row 0 = the shift the kernel chose, rows 1.. = the
    (tiles, 2, N) shifted partial sums"""
    M, K1 = A1.shape
    N = B1.shape[0]
    P1, ldp1, ps1 = x3_planes.planes(B1, False)
    P2, ldp2, ps2 = x3_planes.planes(B2, False)
    buf = torch.empty(1 + 2 * ((M + 63) // 64), N, dtype=torch.float32, device=A1.device)
    bn_shift, part = buf[0], buf[1:]
    _run("hsp_gemm_x3_bn_f32", (_p(A1), _ld(A1), _p(P1), ldp1, ps1, K1, _p(A2), _ld(A2), _p(P2), ldp2, ps2, A2.shape[1], M, N,
                                _p(resid), _ld(resid), _p(cloud_bias), int(rows_per_cloud), _p(out), _ld(out), _p(bn_shift),
                                _p(part), _stream()),
         key=f"M{M}N{N}K{K1}+{A2.shape[1]}bn", abytes=4 * (M * (K1 + A2.shape[1]) + 2 * M * N) + 6 * N * (K1 + A2.shape[1]),
         aflops=2 * M * N * (K1 + A2.shape[1]))
    return buf


def linear_bn_part_ok(x2, W, bias):
    """a Linear (R, K) x (N, K)^T + bias whose product can also leave the first pass of the train-mode BatchNorm behind it
    (``hsp_gemm_x3_bias_bn_f32``)"""
    R, N = x2.shape[0], W.shape[0]
    return (bias is not None and R >= 256 and x2.dtype == torch.float32 and N % 4 == 0 and 256 % (N // 4) == 0
            and gemm_x3_ok(x2, W, None, None, bias, None, None, None, None, R, N) and lib().hsp_gemm_x3_bn_tiles(R, N) <= 512)


def linear_bn_part(x2, W, bias):
    """(x2 W^T + bias,  BatchNorm first-pass buffer (1 + 2 tiles, N): row 0 = the shift (the bias), rows 1.. = (tiles, 2, N)
    shifted column sums of the result) -- the layout ``bn_relu(..., partial=)`` folds"""
    M, K1 = x2.shape
    N = W.shape[0]
    P1, ldp1, ps1 = x3_planes.planes(W, False)
    tiles = lib().hsp_gemm_x3_bn_tiles(M, N)
    out = torch.empty(M, N, dtype=torch.float32, device=x2.device)
    buf = torch.empty(1 + 2 * tiles, N, dtype=torch.float32, device=x2.device)
    _run("hsp_gemm_x3_bias_bn_f32", (_p(x2), _ld(x2), _p(P1), ldp1, ps1, K1, M, N, _p(bias), _p(out), N, _p(buf[0]), _p(buf[1:]),
                                     _stream()),
         key=f"M{M}N{N}K{K1}bn", abytes=4 * (M * K1 + M * N) + 6 * N * K1, aflops=2 * M * N * K1)
    return out, buf


def gemm_x3(A1, B1, nn1=False, A2=None, B2=None, nn2=False, bias=None, resid=None, cloud_bias=None, rows_per_cloud=0, out=None,
            alpha=1.0):
    """the contract of ``gemm_rows`` (fp32 in, fp32 out) with the products formed on the bf16 matrix cores from exact three-way
    bf16 splits of both operands (csrc/gemm_x3.hip): B* are the fp32 weight matrices, split (and transposed where ``nn``) by
    ``x3_planes``"""
    M, K1 = A1.shape
    N = B1.shape[1] if nn1 else B1.shape[0]
    P1, ldp1, ps1 = x3_planes.planes(B1, nn1)
    K2, P2, ldp2, ps2 = 0, None, 0, 0
    if A2 is not None:
        K2 = A2.shape[1]
        P2, ldp2, ps2 = x3_planes.planes(B2, nn2)
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A1.device)
    wsb = lib().hsp_gemm_x3_workspace_bytes(M, N, K1, K2)
    ws = _ws(wsb, A1.device) if wsb else None
    flops = 2 * M * N * (K1 + K2)
    ab = 4 * (M * (K1 + K2) + M * N * (2 if resid is not None else 1)) + 6 * N * (K1 + K2)
    _run("hsp_gemm_x3_f32", (_p(A1), _ld(A1), _p(P1), ldp1, ps1, K1, _p(A2), _ld(A2) if A2 is not None else 0, _p(P2), ldp2, ps2, K2,
                             M, N, _p(bias), _p(resid), _ld(resid) if resid is not None else 0, _p(cloud_bias), int(rows_per_cloud),
                             float(alpha), _p(out), _ld(out), _p(ws), wsb, _stream()),
         key=f"M{M}N{N}K{K1}" + (f"+{K2}" if K2 else ""), abytes=ab, aflops=flops)
    return out


# ------------------------------------------------------------------------------------------------
# dense per-point products of a layer: hand-written kernels only -- csrc/gemm_x3.hip (fp32 products from exact three-way bf16
# splits on the bf16 matrix cores) for every product it covers, csrc/gemm_wave.hip / gemm_rows.hip (fp32 matrix cores) for the
# rest and for the eval-mode forward, the small per-cloud kernels for the 16-row ORL products, csrc/gemm.hip for the parameter
# gradients.  No BLAS-library call.  (The comparison against the tuned BLAS library that bench.py reports lives in
# tools/library_gemm.py.)
# ------------------------------------------------------------------------------------------------
# (tools/library_gemm.py -- bench.py's comparison figure, the tests' second mode -- REPLACES the composites below with BLAS-library
# calls and the ``*_ok`` predicates of the fusions only the hand-written kernels offer with ``False``; nothing in this module
# asks which of the two is running.)


def _layer_out_bn_ok(x2, w_ste, F2, Wa, t2, out3, relu):
    """the layer's out product can also leave the first pass of the BatchNorm behind it (``gemm_x3_bn``)"""
    return (x2.shape[1] != 3 and not relu
            and gemm_x3_ok(x2, w_ste, F2, Wa, None, F2, t2, None, None, x2.shape[0], out3.shape[2])
            and (x2.shape[0] + 63) // 64 <= 512 and out3.shape[1] >= 64)


def _ste_moments_ok(C):
    """the surface layer's (C, 3) STE gradient from the coordinate moments of g (``colsum_rows_xyz``) instead of a product"""
    return C % 4 == 0 and 256 % (C // 4) == 0


def _thin_wgrad_ok(Cout, Cin, R, g):
    """a thin per-point output layer's weight gradient on the split-K kernel through a gradient zero-padded to 64 columns"""
    return Cout < 64 and Cin % 64 == 0 and R >= 1024 and g.is_contiguous()


def _al16(t):
    return t is None or (t.data_ptr() % 16 == 0 and (t.stride(0) * 4) % 16 == 0)


def gemm_own(A1, B1, nn1=False, A2=None, B2=None, nn2=False, bias=None, resid=None, cloud_bias=None, rows_per_cloud=0, out=None,
             alpha=1.0, xyz3=None, w3=None, relu=False):
    """the fp32 product of ``gemm_rows`` on whichever hand-written kernel suits the shape: the LDS-free wave-level kernel
    (csrc/gemm_wave.hip) when the output is large against a short K -- many tiles that each live for a few k-blocks: fm = X W + b
    and the g Wa products; measured 13-14 us against 15-19 us, 24-48 us against 36-67 us -- and the LDS-staged tile kernel
    (csrc/gemm_rows.hip: any K / alignment, split-K) otherwise."""
    M, K1 = A1.shape
    N = B1.shape[1] if nn1 else B1.shape[0]
    K2 = A2.shape[1] if A2 is not None else 0
    if ((M <= 16 or (M <= 64 and K1 % 128 == 0 and (nn1 or all(_al16(t) for t in (A1, B1))))) and A2 is None and bias is None
            and resid is None and cloud_bias is None and xyz3 is None and K1 <= 2048
            and A1.dtype == torch.float32 and A1.stride(1) == 1 and B1.stride(1) == 1):
        return small_rows(A1, B1, nn1, out=out, alpha=alpha)          # a row per cloud: one small launch
    # (the x3 kernel stores -- and reads its residual -- element by element: its output rows need no 16-byte pitch; a result it
    # allocates itself for N = 1286 has none either)
    if M >= 256 and gemm_x3_ok(A1, B1, A2, B2, bias, resid, cloud_bias, xyz3, out, M, N) and (out is None or out.stride(1) == 1):
        res = gemm_x3(A1, B1, nn1, A2, B2, nn2, bias=bias, resid=resid, cloud_bias=cloud_bias, rows_per_cloud=rows_per_cloud,
                      out=out, alpha=alpha)
        return torch.relu_(res) if relu else res
    two, rc = A2 is not None, resid is not None and cloud_bias is not None
    plain = bias is None and resid is None and cloud_bias is None and xyz3 is None
    # the forms gemm_wave.hip instantiates: fm (nn + bias), g W (nn), x W^T (nt [+ bias]), out (nt + nt, residual + cloud bias),
    # surface out (nt, residual + cloud bias + xyz3), gX (nn + nt)
    form = ((not two and nn1 and (plain or (bias is not None and resid is None and cloud_bias is None and xyz3 is None)))
            or (not two and not nn1 and (plain or (bias is not None and resid is None and cloud_bias is None and xyz3 is None)))
            or (two and not nn1 and not nn2 and bias is None and rc and xyz3 is None)
            or (not two and not nn1 and bias is None and rc and xyz3 is not None)
            or (two and nn1 and not nn2 and plain))
    if (form and A1.dtype == torch.float32 and K1 + K2 <= 512 and M * N >= 512 * 1024 and K1 % 32 == 0 and K2 % 32 == 0
            and N % 32 == 0 and (not rc or rows_per_cloud >= 64) and all(_al16(t) for t in (A1, B1, A2, B2, resid, out))):
        return gemm_wave(A1, B1, nn1, A2, B2, nn2, bias=bias, resid=resid, cloud_bias=cloud_bias, rows_per_cloud=rows_per_cloud,
                         out=out, alpha=alpha, xyz3=xyz3, w3=w3, cfg=(1 << 29) if relu else 0)
    res = gemm_rows(A1, B1, nn1, A2, B2, nn2, bias=bias, resid=resid, cloud_bias=cloud_bias, rows_per_cloud=rows_per_cloud,
                    out=out, alpha=alpha, xyz3=xyz3, w3=w3)
    return torch.relu_(res) if relu else res


def _fm_rows(X2, weights, bias, out=None):
    """fm = X W + b   (gcn3d.py:171)"""
    return gemm_own(X2, weights, True, bias=bias, out=out)


def _layer_out_rows(x2, w_ste, F2, Wa, t2, out3, relu=False, bn_shift=None):
    """... ``bn_shift`` not None (own mode, x3 shapes): also returns the BatchNorm first-pass buffer of the result (else None)"""
    if bn_shift is not None and _layer_out_bn_ok(x2, w_ste, F2, Wa, t2, out3, relu):
        B_, N_, C_ = out3.shape
        return gemm_x3_bn(x2, w_ste, F2, Wa, F2, t2.contiguous(), N_, out3.view(B_ * N_, C_))
    _layer_out_rows_plain(x2, w_ste, F2, Wa, t2, out3, relu)
    return None


def _layer_out_rows_plain(x2, w_ste, F2, Wa, t2, out3, relu=False):
    """out = x Wste^T + F Wa^T + F + t[cloud]   (gcn3d.py:149,186,156): one fused launch"""
    B, N, C = out3.shape
    out = out3.view(B * N, C)
    if x2.shape[1] == 3:                                  # HSlayer_surface: the K = 3 STE on raw coordinates rides in the epilogue
        gemm_own(F2, Wa, False, resid=F2, cloud_bias=t2.contiguous(), rows_per_cloud=N, out=out, xyz3=x2,
                 w3=w_ste.contiguous(), relu=relu)       # (... and so does the relu that follows conv_0)
    else:
        gemm_own(x2, w_ste, False, F2, Wa, False, resid=F2, cloud_bias=t2.contiguous(), rows_per_cloud=N, out=out)
        if relu:
            torch.relu_(out3)
    return out3


def _mm_nn(g2, W, out=None, alpha=1.0):
    """alpha * (g @ W) for a row-strided (K,N) matrix W"""
    return gemm_own(g2, W, True, out=out, alpha=alpha)


def _mm_nt(x2, W, bias=None, out=None):
    """x @ W^T (+ bias) for a (N,K) weight"""
    return gemm_own(x2, W, False, bias=bias, out=out)


def _grad_in_rows(g2, w_ste, gfm2, weights, out):
    """gX = g Wste + gfm W^T   (input gradient of gcn3d.py:149 and :171)"""
    return gemm_own(g2, w_ste, True, gfm2, weights, False, out=out)


def small_rows(A, W, nn=False, out=None, alpha=1.0):
    """out (M,N) = alpha * A (M,K) op(W) for M <= 16 per-cloud rows (csrc/gemm_x3.hip, one launch): W (N,K), or (K,N) with ``nn``"""
    M, K = A.shape
    N = W.shape[1] if nn else W.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    _run("hsp_small_rows_f32", (_p(A), _ld(A), _p(W), _ld(W), 1 if nn else 0, M, N, K, float(alpha), _p(out), _ld(out), _stream()),
         key=f"M{M}N{N}K{K}{'nn' if nn else 'nt'}", abytes=4 * (M * K + N * K + M * N))
    return out


def _tiny_tn(a, b, out, mom=None, gste=None):
    """out = a^T b for per-cloud rows a (B,M), b (B,N) (B = 16): one small launch (hsp_small_outer_f32); ``mom`` / ``gste``: the
    surface layer's (C,3) STE gradient as a rider (the sum over the clouds of the coordinate moments of g)"""
    B, Ma = a.shape
    Cm = mom.shape[1] // 4 if mom is not None else 0
    if B > 64:
        # hsp_small_outer_f32 holds one row per lane (B <= 64 clouds).  Larger per-GPU batches: the general weight-gradient
        # path for a^T b (any row count), the (C, 3) coordinate-moment rider as a plain column sum over the clouds
        wgrad(a.contiguous(), b.contiguous(), out=out)
        if mom is not None:
            gste.copy_(mom[:, Cm:].sum(dim=0).view(3, Cm).t())
        return out
    _run("hsp_small_outer_f32", (_p(a), _ld(a), _p(b), _ld(b), B, Ma, b.shape[1], _p(out), _ld(out),
                                 _p(mom[:, Cm:]) if mom is not None else None, _ld(mom) if mom is not None else 0, Cm,
                                 _p(gste), _stream()),
         key=f"B{B}M{Ma}N{b.shape[1]}", abytes=4 * (B * (Ma + b.shape[1]) + Ma * b.shape[1]))
    return out


def _orl_fwd_raw(F3, idx_x, k):
    """(fg (B,C), argmax (B,N,C) uint8): mean over points of the neighbourhood max, one pass, no (B,N,C) max tensor"""
    B, N, C = F3.shape
    fg = torch.empty(B, C, dtype=torch.float32, device=F3.device)
    arg = torch.empty(B, N, C, dtype=torch.uint8, device=F3.device)
    L = lib()
    wsb = L.hsp_orl_workspace_bytes(B, N, C)
    ws = _ws(wsb, F3.device)
    _run("hsp_orl_global_fwd", (_p(F3), _p(idx_x), B, N, k, idx_x.shape[2], C, _p(fg), _p(arg), _p(ws), wsb, _stream()),
         key=f"B{B}N{N}k{k}C{C}", abytes=B * N * (4 * C + 4 * k + C))
    return fg, arg


def _residual_bias(out3, f3, t2):
    """out3 (B,N,C) += f3 + t2[:, None, :]  in place"""
    B, N, C = out3.shape
    _run("hsp_residual_bias", (_p(out3), _p(f3), _p(t2.contiguous()), B, N, C, _stream()), key=f"B{B}N{N}C{C}",
         abytes=12 * B * N * C)


def colsum_rows(x3):
    """(B,C) = x3.sum(dim=1) for a contiguous (B,N,C) fp32 tensor: deterministic two-stage column sum"""
    B, N, C = x3.shape
    if ((C % 4 or 256 % (C // 4)) and C > 256) or not x3.is_contiguous():
        return x3.sum(dim=1)
    out = torch.empty(B, C, dtype=torch.float32, device=x3.device)
    L = lib()
    wsb = L.hsp_orl_workspace_bytes(B, N, C)
    ws = _ws(wsb, x3.device)
    _run("hsp_colsum_rows", (_p(x3), B, N, C, _p(out), _p(ws), wsb, _stream()), key=f"B{B}N{N}C{C}", abytes=4 * B * N * C)
    return out


def colsum_rows_xyz(g, xyz):
    """(B, 4C) fp32: per-cloud column sums of g (B,N,C) fp32 / bf16 in [:, :C] and its three coordinate moments
    sum_i g[b,i,:] * xyz[b,i,j] in [:, (1+j)C:(2+j)C] -- one pass (the surface layer's gt and its STE weight gradient)"""
    B, N, C = g.shape
    mom = torch.empty(B, 4 * C, dtype=torch.float32, device=g.device)
    wsb = 4 * lib().hsp_orl_workspace_bytes(B, N, C)
    ws = _ws(wsb, g.device)
    _run("hsp_colsum_rows_xyz" + _sfx(g), (_p(g), _p(xyz), B, N, C, _p(mom), _p(ws), wsb, _stream()), key=f"B{B}N{N}C{C}",
         abytes=B * N * (_es(g) * C + 12))
    return mom


def _orl_bwd_accumulate_raw(gfg_over_n, idx_x, arg, k, gF3, extra=None):
    """gF3[b,m,c] += extra[b,m,c] + gfg_over_n[b,c] * #{i : idx_x[b,i,arg[b,i,c]] == m}     (in place, one pass)"""
    B, N, C = gF3.shape
    if DETERMINISTIC:
        tmp = torch.empty_like(gF3)
        off, edge = rev_index(idx_x, k, N)
        _run("hsp_gather_max_bwd_csr", (_p(gfg_over_n), 1, _p(arg), _p(off), _p(edge), B, N, N, k, C, _p(tmp), _stream()),
             key=f"B{B}Ns{N}Nq{N}k{k}C{C}bc", abytes=B * N * (4 * C + 8 * k + C))
        gF3.add_(tmp)
        if extra is not None:
            gF3.add_(extra)
    else:
        _run("hsp_gather_max_bwd", (_p(gfg_over_n), 1, _p(idx_x), _vp(0), _p(arg), B, N, N, N, idx_x.shape[2], C,
                                    _p(gF3), 1, _p(extra), _stream()),
             key=f"B{B}Ns{N}Nq{N}C{C}bc+", abytes=B * N * (12 * C + 4 * idx_x.shape[2] + C))


def _rf_conv_fwd_raw(xyz, idx, directions, fm, S, need_bwd=True):
    """(out (B,N,C), argrow (B,N,SC) uint16, fwin (B,N,SC) | None).  fwin = the winners' support values, all the
    column-tile backward needs of fm once a cloud's fm outgrows L2 (hsp_rf_conv_wants_fwin); smaller layers and
    the gather-form backward (DETERMINISTIC) read fm itself."""
    B, N, k = idx.shape
    SC = directions.shape[1]
    C = SC // S
    out = torch.empty(B, N, C, dtype=torch.float32, device=xyz.device)
    arg = torch.empty(B, N, SC, dtype=torch.uint16, device=xyz.device)
    want = need_bwd and not DETERMINISTIC and lib().hsp_rf_conv_wants_fwin(N, S, C)
    fwin = torch.empty(B, N, SC, dtype=torch.float32, device=xyz.device) if want else None
    key = f"B{B}N{N}k{k}S{S}C{C}"
    # algorithmic bytes per point: (S+1) C 4 + 4 k + 12 in, 4 C + S C out (a one-byte arg-max slot).  The
    # kernel itself writes a uint16 winning ROW per slot (the backward then needs neither idx nor the slot -> one gather fewer) and,
    # when a cloud's fm outgrows L2, the winners' support values: its designed stream is recorded next to the algorithmic count
    design_stream_bytes[("hsp_rf_conv_fwd", key)] = B * N * (12 + 4 * k + 4 * (S + 1) * C + 4 * C + 2 * SC
                                                             + (4 * SC if fwin is not None else 0)) + 12 * SC
    _run("hsp_rf_conv_fwd", (_p(xyz), _p(idx), _p(directions), _p(fm), B, N, k, S, C, _p(out), _p(arg), _p(fwin),
                             _stream()),
         key=key, abytes=B * N * (12 + 4 * k + 4 * (S + 1) * C + 4 * C + SC) + 12 * SC)
    return out, arg, fwin


def _rf_conv_bwd_raw(xyz, idx, directions, fm, arg, gF3, S):
    """fm: either fm (B,N,(S+1)C) or the forward's fwin (B,N,SC) (told apart by the width)."""
    B, N, k = idx.shape
    SC = directions.shape[1]
    C = SC // S
    gfm = torch.empty(B, N, (S + 1) * C, dtype=torch.float32, device=gF3.device)
    gd = torch.empty_like(directions)
    L = lib()
    if DETERMINISTIC:
        wsb = L.hsp_rf_bwd_workspace_bytes(SC)
        ws = _ws(wsb, gF3.device)
        off, edge = rev_index(idx, k, N)
        # (section 4: S C + 4 S C + 4 C in -- arg slot, the winners' support values, grad_out -- and 4 (S+1) C out per point)
        design_stream_bytes[("hsp_rf_conv_bwd", f"B{B}N{N}k{k}S{S}C{C}")] = B * N * (12 + 8 * k + 6 * SC + 4 * C + 4 * (S + 1) * C) + 24 * SC
        _run("hsp_rf_conv_bwd", (_p(xyz), _p(directions), _p(fm), _p(arg), _p(gF3), _p(off), _p(edge), B, N, k, S, C,
                                 _p(gfm), _p(gd), _p(ws), wsb, _stream()),
             key=f"B{B}N{N}k{k}S{S}C{C}", abytes=B * N * (12 + 8 * k + 5 * SC + 4 * C + 4 * (S + 1) * C) + 24 * SC)
    else:
        wsb = L.hsp_rf_bwd_scatter_workspace_bytes(B, SC)
        ws = _ws(wsb, gF3.device)
        is_fwin = fm.shape[-1] == SC
        design_stream_bytes[("hsp_rf_conv_bwd_scatter", f"B{B}N{N}S{S}C{C}")] = B * N * (12 + 6 * SC + 4 * C + 4 * (S + 1) * C) + 24 * SC
        _rf_bwd_dirs_call("hsp_rf_conv_bwd_scatter", (_p(xyz), _p(directions), _p(None if is_fwin else fm), _p(fm if is_fwin else None),
                                                      _p(arg), _p(gF3), B, N, S, C, _p(gfm), _p(gd)), ws, wsb, (directions, gd),
                          key=f"B{B}N{N}S{S}C{C}", abytes=B * N * (12 + 5 * SC + 4 * C + 4 * (S + 1) * C) + 24 * SC)
    return gfm, gd


# ------------------------------------------------------------------------------------------------
# eval-mode forward in the reference's arithmetic
# The feature-space neighbour search ranks rows by distances that cancel catastrophically: two forwards that differ in the
# last bit of a feature row pick different neighbours for a few rows per thousand, and the outputs then differ by 1e-3, not
# 1e-7.  The reference's CPU forward is reproducible operation by operation (oracle/gen_golden_exact.py pins each statement
# against the imported reference): its GEMMs for K <= 256 are k-ordered fp32 fma chains from 0 (K = 512: two such chains added),
# mean over the supports is a sequential sum / S, mean over the points is ATen's 16-row cascade / N, eval-mode BatchNorm is
# ((x - m) * invstd) * w + b.  With ``exact_scope(True)`` (FaceRecon sets it whenever the module is in eval mode) the HS layers
# run exactly those orders: csrc/gemm_wave.hip's products ARE that chain (v_mfma_f32_32x32x2_f32, ascending k),
# hsp_layer_out_exact_f32 adds in the reference's order, hsp_orl_global_exact_f32 is the cascade, hsp_bn_eval_f32 the BatchNorm.
# The training forward keeps the faster forms (fp32 from bf16 splits): its BatchNorm batch statistics, dropout and
# device-side augmentation draws rule bit-level agreement out anyway.
# ------------------------------------------------------------------------------------------------
_exact = False


class exact_scope:
    """within the scope the HS layers' forward runs in the reference's operation order (see above)"""

    def __init__(self, on=True):
        self.on = bool(on) and os.environ.get("HSP_EXACT", "1") != "0"

    def __enter__(self):
        global _exact
        self.prev, _exact = _exact, self.on
        return self

    def __exit__(self, *exc):
        global _exact
        _exact = self.prev
        return False


def exact_forward():
    return _exact


def _ext_inference():
    """True when an eval-mode forward may take the C++ binding's one-call forms (csrc/hsp_torch.cpp): exact scope, no autograd
    graph to record, no per-call event timer attached (bench.py --breakdown times the ctypes calls)"""
    return _exact and _timer is None and not torch.is_grad_enabled() and _ext_ok()


def _f32c(*ts):
    return all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in ts)


def _exact_layer_ok(N, Cin, C, tensors):
    """shapes the exact forms cover: channel counts that are multiples of 32, conv2 over at most 512 input channels (one chain
    or two; conv_4's 1024 are four blocks and its rows rank nothing), clouds of at least 32 points, 16-byte aligned rows"""
    return (C % 32 == 0 and (Cin == 3 or Cin % 32 == 0) and 2 * C <= 512 and N >= 32
            and all(t is None or (t.dtype == torch.float32 and _al16(t)) for t in tensors))


def _orl_fwd_exact(F3, idx_x, k):
    """_orl_fwd_raw with the reference's summation order: (fg (B,C), argmax (B,N,C) uint8)"""
    B, N, C = F3.shape
    fg = torch.empty(B, C, dtype=torch.float32, device=F3.device)
    arg = torch.empty(B, N, C, dtype=torch.uint8, device=F3.device)
    wsb = lib().hsp_orl_exact_workspace_bytes(B, N, C)
    ws = _ws(wsb, F3.device)
    _run("hsp_orl_global_exact_f32", (_p(F3), _p(idx_x), B, N, k, idx_x.shape[2], C, _p(fg), _p(arg), _p(ws), wsb, _stream()),
         key=f"B{B}N{N}k{k}C{C}", abytes=B * N * (4 * C + 4 * k + C))
    return fg, arg


def _layer_out_exact(F2, w_conv2, fg, N, out3, ste=None, xyz3=None, w3=None, relu=False):
    """out = ((conv2(cat[F, f_global])) + F) + STE in the reference's order (hsp_layer_out_exact_f32)"""
    R, C = F2.shape
    Wa, Wb = w_conv2[:, :C], w_conv2[:, C:]
    out = out3.view(R, C)
    two = 2 * C > 256
    t2 = gemm_wave(fg, Wb, False) if two else None             # K = 512: the f_global block is its own chain
    _run("hsp_layer_out_exact_f32", (_p(F2), _ld(F2), _p(Wa), _ld(Wa), _p(None if two else fg), 0 if two else _ld(fg),
                                     _p(None if two else Wb), 0 if two else _ld(Wb), _p(t2), 1 if two else 0,
                                     _p(ste), _ld(ste) if ste is not None else 0, _p(xyz3), _p(w3), 1 if relu else 0, R, C, N,
                                     _p(out), _ld(out), _stream()),
         key=f"R{R}C{C}{'x' if xyz3 is not None else ''}", abytes=4 * R * C * 4, aflops=2 * R * C * C * (1 if two else 2))
    return out3


class _HSLayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, X, idx_f, idx_x, k, S, weights, bias, directions, w_ste3, w_conv23, bn_shift=None):
        # Conv1d weights arrive in their native (out, in, 1) shape and their gradients are returned in it:
        # a squeezed view would make AccumulateGrad clone every gradient (one D2D copy per tensor and step)
        w_ste, w_conv2 = w_ste3.squeeze(-1), w_conv23.squeeze(-1)
        xyz = _req(xyz, torch.float32, "hs_layer.xyz")
        X = _req(X, torch.float32, "hs_layer.X")
        idx_f = _req(idx_f, torch.int32, "hs_layer.idx_f")
        idx_x = _req(idx_x, torch.int32, "hs_layer.idx_x")
        directions = _req(directions, torch.float32, "hs_layer.directions")
        B, N, Cin = X.shape
        SC = directions.shape[1]
        C = SC // S
        X2 = X.view(B * N, Cin)
        exact = _exact and bn_shift is None and _exact_layer_ok(N, Cin, C, (X2, weights, w_ste, w_conv2))
        # exact: fm = (X W, a k-ordered chain) + b -- gcn3d.py:171 as the reference's CPU GEMM rounds it
        fm = gemm_wave(X2, weights, True, bias=bias) if exact else _fm_rows(X2, weights, bias)      # (BN, (S+1)C)
        need_bwd = any(ctx.needs_input_grad)
        F3, arg, fwin = _rf_conv_fwd_raw(xyz, idx_f, directions, fm.view(B, N, -1), S, need_bwd)
        if fwin is not None:
            fm = fwin                                                          # fm itself is no longer needed
        fm = fm.view(B, N, -1)
        F2 = F3.view(B * N, C)
        out3 = torch.empty(B, N, C, dtype=torch.float32, device=X.device)      # (returned as is: not a view)
        if exact:
            fg, arg_o = _orl_fwd_exact(F3, idx_x, k)
            _layer_out_exact(F2, w_conv2, fg, N, out3, ste=gemm_wave(X2, w_ste, False))
            part = None
        else:
            fg, arg_o = _orl_fwd_raw(F3, idx_x, k)                             # (B,C)
            t2 = _mm_nt(fg, w_conv2[:, C:])                                    # (B,C): the per-cloud half of conv2
            part = _layer_out_rows(X2, w_ste, F2, w_conv2[:, :C], t2, out3, bn_shift=bn_shift)   # X Wste^T + F Wa^T + F + t[b]
        ctx.save_for_backward(xyz, X, idx_f, idx_x, fm, arg, F3, arg_o, fg, weights, directions, w_ste3, w_conv23)
        ctx.k, ctx.S, ctx.x3 = k, S, x3_planes
        ctx.with_part = bn_shift is not None
        if bn_shift is not None:
            # second output: the BatchNorm partial sums of out3 (or an empty tensor when the product that ran does not leave
            # them); not differentiable
            if part is None:
                part = torch.empty(0, dtype=torch.float32, device=out3.device)
            ctx.mark_non_differentiable(part)
            ctx.set_materialize_grads(False)
            return out3, part
        return out3

    @staticmethod
    def backward(ctx, g, _gpart=None):
        xyz, X, idx_f, idx_x, fm, arg, F3, arg_o, fg, weights, directions, w_ste3, w_conv23 = ctx.saved_tensors
        w_ste, w_conv2 = w_ste3.squeeze(-1), w_conv23.squeeze(-1)
        k, S = ctx.k, ctx.S
        g = _req(g, torch.float32, "hs_layer.grad")
        B, N, Cin = X.shape
        C = F3.shape[2]
        g2, X2, F2 = g.view(B * N, C), X.view(B * N, Cin), F3.view(B * N, C)
        Wa, Wb = w_conv2[:, :C], w_conv2[:, C:]
        gt = colsum_rows(g)                                                    # (B,C) = sum_i g
        g_conv2 = torch.empty_like(w_conv2)
        with WgradBatch(), x3_scope(ctx.x3):                                   # the three parameter gradients: one fold launch
            g_ste = torch.empty(C, Cin, dtype=torch.float32, device=g.device)
            wgrad_pair(g2, F2, g_conv2[:, :C], g2, X2, g_ste)                  # gWa (in place, ldc = 2C) and gWste: one split-K launch
            _tiny_tn(gt, fg, g_conv2[:, C:])                                   # gWb = gt^T fg (tiny), straight into its column block
            gF3 = torch.empty(B, N, C, dtype=torch.float32, device=g.device)
            _mm_nn(g2, Wa, out=gF3.view(B * N, C))                             # g Wa ...
            _orl_bwd_accumulate_raw(_mm_nn(gt, Wb, alpha=1.0 / N), idx_x, arg_o, k, gF3, extra=g)  # ... + g + ORL scatter, one pass
            gfm, gD = _rf_conv_bwd_raw(xyz, idx_f, directions, fm.view(B, N, -1), arg, gF3, S)
            gfm2 = gfm.view(B * N, -1)
            gW, gb = wgrad(X2, gfm2, colsum=True)                              # X^T gfm and the bias gradient
            gX3 = torch.empty(B, N, Cin, dtype=torch.float32, device=g.device)
            _grad_in_rows(g2, w_ste, gfm2, weights, gX3.view(B * N, Cin))      # g Wste + gfm W^T
        return None, gX3, None, None, None, None, gW, gb, gD, g_ste.unsqueeze_(-1), g_conv2.unsqueeze_(-1), None


class _SurfaceLayer(torch.autograd.Function):
    """HSlayer_surface.forward (gcn3d.py:79-90) as one node; xyz carries no gradient."""

    @staticmethod
    def forward(ctx, xyz, idx_x, k, S, directions, w_ste3, w_conv23, relu=False):
        w_ste, w_conv2 = w_ste3.squeeze(-1), w_conv23.squeeze(-1)
        xyz = _req(xyz, torch.float32, "surface_layer.xyz")
        idx_x = _req(idx_x, torch.int32, "surface_layer.idx")
        directions = _req(directions, torch.float32, "surface_layer.directions")
        B, N, _ = xyz.shape
        SC = directions.shape[1]
        C = SC // S
        if idx_x.shape[2] != k:
            raise HspError("surface_layer: idx must have exactly k columns")
        F3 = torch.empty(B, N, C, dtype=torch.float32, device=xyz.device)
        arg = torch.empty(B, N, SC, dtype=torch.uint16, device=xyz.device)
        _run("hsp_rf_surface_fwd", (_p(xyz), _p(idx_x), _p(directions), B, N, k, S, C, _p(F3), _p(arg), _stream()),
             key=f"B{B}N{N}k{k}S{S}C{C}", abytes=B * N * (12 + 4 * k + 4 * C + SC) + 12 * SC)
        F2, x2 = F3.view(B * N, C), xyz.view(B * N, 3)
        out3 = torch.empty(B, N, C, dtype=torch.float32, device=xyz.device)
        if _exact and _exact_layer_ok(N, 3, C, (w_conv2,)):
            fg, arg_o = _orl_fwd_exact(F3, idx_x, k)
            _layer_out_exact(F2, w_conv2, fg, N, out3, xyz3=x2, w3=w_ste.contiguous(), relu=relu)
        else:
            fg, arg_o = _orl_fwd_raw(F3, idx_x, k)
            t2 = _mm_nt(fg, w_conv2[:, C:])
            _layer_out_rows(x2, w_ste, F2, w_conv2[:, :C], t2, out3, relu=relu)
        ctx.k, ctx.S, ctx.relu, ctx.x3 = k, S, relu, x3_planes
        if relu:
            # relu(conv_0(...)) (FaceRecon.py:88) inside the node: the relu rides in the product's epilogue, the result is handed
            # out TWICE (conv_1 and the concat read it) and the two gradients + the relu mask meet in ONE pass in backward
            ctx.save_for_backward(xyz, idx_x, arg, F3, arg_o, fg, directions, w_conv23, out3)
            ctx.set_materialize_grads(False)
            return out3, out3.view_as(out3)
        ctx.save_for_backward(xyz, idx_x, arg, F3, arg_o, fg, directions, w_conv23)
        return out3

    @staticmethod
    def backward(ctx, *gs):
        if ctx.relu:
            xyz, idx_x, arg, F3, arg_o, fg, directions, w_conv23, y = ctx.saved_tensors
            gs = [t for t in gs if t is not None]
            if not gs:
                return (None,) * 8
            B, N, C = F3.shape
            (ga, lda) = _rows_pitch(gs[0], C)
            (gb, ldb) = _rows_pitch(gs[1], C) if len(gs) > 1 else (None, 0)
            g = torch.empty(B, N, C, dtype=torch.float32, device=y.device)
            _run("hsp_add_relu_bwd", (_p(ga), lda, _p(gb), ldb, _p(y), B * N, C, _p(g), _stream()), key=f"R{B * N}C{C}",
                 abytes=4 * (2 + len(gs)) * B * N * C)
        else:
            xyz, idx_x, arg, F3, arg_o, fg, directions, w_conv23 = ctx.saved_tensors
            g = gs[0]
        w_conv2 = w_conv23.squeeze(-1)
        k, S = ctx.k, ctx.S
        g = _req(g, torch.float32, "surface_layer.grad")
        B, N, C = F3.shape
        SC = directions.shape[1]
        g2, F2, x2 = g.view(B * N, C), F3.view(B * N, C), xyz.view(B * N, 3)
        Wa, Wb = w_conv2[:, :C], w_conv2[:, C:]
        g_conv2 = torch.empty_like(w_conv2)
        own_ste = _ste_moments_ok(C)
        if own_ste:
            # gt = sum_i g and the per-cloud coordinate moments of g in one pass; the STE gradient g^T xyz is their sum over the
            # batch, taken as a rider of the gt^T fg launch: no library GEMM for the (C, 3) product
            mom = colsum_rows_xyz(g, xyz)
            gt = mom[:, :C]
            g_ste = torch.empty(C, 3, dtype=torch.float32, device=g.device)
        else:
            gt = colsum_rows(g)
        with WgradBatch():                                  # (its fold goes out with the step's folds inside a StepFolds scope)
            wgrad(g2, F2, out=g_conv2[:, :C])
        if own_ste:
            _tiny_tn(gt, fg, g_conv2[:, C:], mom=mom, gste=g_ste)
        else:
            _tiny_tn(gt, fg, g_conv2[:, C:])
        gF3 = torch.empty(B, N, C, dtype=torch.float32, device=g.device)
        with x3_scope(ctx.x3):
            _mm_nn(g2, Wa, out=gF3.view(B * N, C))
        _orl_bwd_accumulate_raw(_mm_nn(gt, Wb, alpha=1.0 / N), idx_x, arg_o, k, gF3, extra=g)
        gD = torch.empty_like(directions)
        L = lib()
        wsb = L.hsp_rf_bwd_scatter_workspace_bytes(B, SC)
        ws = _ws(wsb, g.device)
        _rf_bwd_dirs_call("hsp_rf_surface_bwd", (_p(xyz), _p(directions), _p(arg), _p(gF3), B, N, S, C, _p(gD)), ws, wsb,
                          (directions, gD), key=f"B{B}N{N}S{S}C{C}", abytes=B * N * (12 + 4 * C + SC) + 24 * SC)
        if not own_ste:
            g_ste = wgrad(g2, x2)
        return None, None, None, None, gD, g_ste.unsqueeze_(-1), g_conv2.unsqueeze_(-1), None


def hs_layer(xyz, X, idx_f, idx_x, k, S, weights, bias, directions, w_ste, w_conv2, bn_shift=None):
    """HS_layer.forward (gcn3d.py:143-156) given the feature-space (idx_f, exactly k columns) and xyz-space
    (idx_x, >= k columns) neighbour indices; w_ste (Cout,Cin,1), w_conv2 (Cout,2*Cout,1): the Conv1d weights.
    ``bn_shift`` not None (a train-mode BatchNorm follows): returns (out, partial) where ``partial`` holds the first pass of that
    BatchNorm's statistics, left by the out product's epilogue (empty when that product did not run on the kernel that does
    it); pass both to ``bn_relu(out, bn, partial=partial)``."""
    if bn_shift is None:
        if (_ext_inference() and _f32c(xyz, X, weights, bias, directions, w_ste, w_conv2) and idx_f.dtype == torch.int32
                and idx_x.dtype == torch.int32 and idx_f.is_contiguous() and idx_x.is_contiguous()
                and _exact_layer_ok(X.shape[1], X.shape[2], directions.shape[1] // S, (X, weights))):
            from ._ext import ext                      # inference: the layer's launch sequence issued from C++ in one call
            return ext().hs_layer_forward(xyz, X, idx_f, idx_x, k, S, weights, bias, directions, w_ste, w_conv2)
        return _HSLayer.apply(xyz, X, idx_f, idx_x, k, S, weights, bias, directions, w_ste, w_conv2)
    return _HSLayer.apply(xyz, X, idx_f, idx_x, k, S, weights, bias, directions, w_ste, w_conv2, bn_shift)


def surface_layer(xyz, idx_x, k, S, directions, w_ste, w_conv2, relu=False):
    """HSlayer_surface.forward (gcn3d.py:79-90) given the xyz neighbour index (exactly k columns).  ``relu``: apply the relu
    that follows the layer (FaceRecon.py:88) inside the node and return the result TWICE (one tensor per consumer)."""
    if (_ext_inference() and _f32c(xyz, directions, w_ste, w_conv2) and idx_x.dtype == torch.int32 and idx_x.is_contiguous()
            and idx_x.shape[2] == k and _exact_layer_ok(xyz.shape[1], 3, directions.shape[1] // S, (w_conv2.squeeze(-1),))):
        from ._ext import ext
        y = ext().surface_layer_forward(xyz, idx_x, k, S, directions, w_ste, w_conv2, relu)
        return (y, y.view_as(y)) if relu else y
    return _SurfaceLayer.apply(xyz, idx_x, k, S, directions, w_ste, w_conv2, relu)


class _LinearRows(torch.autograd.Function):
    """y = x W^T + b over point rows (the Conv1d(k=1) layers of the heads, PoseR.py:27-36 / PoseTs.py:32-44 /
    FaceRecon.py:38-66): library GEMMs for y and dx, the parameter-gradient kernel for dW with the bias gradient as its
    fused column sum (or the two-stage column-sum kernel) -- no ATen multi-block reduction anywhere, which keeps the node
    replayable inside a hipGraph (graph.py::GraphedNetwork)."""

    @staticmethod
    def forward(ctx, x2, weight, bias, want_part=False):
        if x2.dim() == 2 and x2.stride(1) == 1 and x2.stride(0) >= x2.shape[1] and x2.dtype == torch.float32 and x2.is_cuda:
            pass                                    # rows of a wider buffer (feat's padded pitch): every consumer takes a row stride
        else:
            x2 = _req(x2, torch.float32, "linear_rows.x")
        ctx.save_for_backward(x2, weight)
        ctx.has_bias = bias is not None
        ctx.x3 = x3_planes
        if want_part:
            # second output: the first pass of the BatchNorm that follows (or an empty tensor when this product does not leave
            # it); not differentiable
            if linear_bn_part_ok(x2, weight, bias):
                y, part = linear_bn_part(x2, weight, bias)
            else:
                y, part = _mm_nt(x2, weight, bias), torch.empty(0, dtype=torch.float32, device=x2.device)
            ctx.mark_non_differentiable(part)
            ctx.set_materialize_grads(False)
            return y, part
        return _mm_nt(x2, weight, bias)

    @staticmethod
    def backward(ctx, g, _gpart=None):
        x2, weight = ctx.saved_tensors
        g = _req(g, torch.float32, "linear_rows.grad")
        with x3_scope(ctx.x3):
            gx = _mm_nn(g, weight) if ctx.needs_input_grad[0] else None
        R, Cout = g.shape
        Cin = x2.shape[1]
        gb = None
        if (Cout % 64 == 0 and Cin % 64 == 0) or _wgrad_ragged_ok(x2, g, g):
            gwt, gb = wgrad(x2, g, colsum=True)                 # (Cin, Cout) = dW^T, column sums of g = db
            gw = gwt.t()
        elif _thin_wgrad_ok(Cout, Cin, R, g):
            # a thin per-point output layer (the 3- and 30-wide last layers of FaceRecon.py:48,68 over B*N rows): g padded to
            # 64 zero columns takes the split-K kernel -- the library's 16448-deep 30 x 128 product is a 100 us launch
            gp = g.new_zeros(R, 64)
            gp[:, :Cout] = g
            gwt, gbp = wgrad(x2, gp, colsum=True)
            gw, gb = gwt[:, :Cout].t(), gbp[:Cout]
        else:
            if R <= 64 and g.stride(1) == 1 and x2.stride(1) == 1:
                gw = _tiny_tn(g, x2, torch.empty(Cout, Cin, dtype=torch.float32, device=g.device))   # per-cloud rows (the towers' conv4)
            else:
                gw = wgrad(g, x2)
            if ctx.has_bias:
                gb = colsum_rows(g.view(1, R, Cout)).view(Cout)
        return gx, gw, (gb if ctx.has_bias else None), None


class _FanGroup:
    """the consumers of one set of rows (one forward pass): the input-gradient buffer they accumulate into"""

    def __init__(self):
        self.gx = None
        self.members = 0             # set by fan_linear_rows
        self.seen = 0                # members whose backward has run in the current pass

    def end_of_pass(self):
        self.gx, self.seen = None, 0


class _FanMember(torch.autograd.Function):
    """One of several Linear / Conv1d(k=1) layers fed by the SAME rows x (the first layers of the three pose heads and of the
    reconstruction block all read ``feat``: PoseR.py:27, PoseTs.py:32 on cat[feat, xyz], FaceRecon.py:38).  Forward: the plain
    product.  Backward: the layers' input gradients are ONE accumulation chain  gx = g_0 W_0;  gx = g_1 W_1 + gx;  ...  carried in
    the products' epilogues (residual read and written in place) -- the member whose backward runs first allocates the buffer
    and RETURNS it as the gradient of x, the others add into it in place and return None (= zero): autograd then has nothing to
    sum (three 250 MB element-wise passes at B=16, N=1028), and every member still runs when ITS upstream gradient arrives,
    while that gradient is still cache-resident (one node for all four was measured 0.2 ms SLOWER per step: it runs when the
    last gradient arrives, after 270 MB of them have gone through a 256 MB cache).  Stream order makes the buffer complete
    before x's producer reads it (the members share an alias of x -- ``fan_linear_rows`` -- so a gradient from a consumer
    outside the group is only added once the group's buffer is final).  The layer on cat[x, xyz] (``xw``: the pitched concatenation) contributes through the first K
    columns of its weight; the coordinates carry no gradient.  One backward pass per forward (no retain_graph)."""

    @staticmethod
    def forward(ctx, x, xw, group, w, b):
        src = x if xw is None else xw
        if linear_bn_part_ok(src, w, b):              # (every member is followed by a BatchNorm: its first pass rides along)
            y, part = linear_bn_part(src, w, b)
        else:
            y, part = gemm_own(src, w, False, bias=b), torch.empty(0, dtype=torch.float32, device=x.device)
        ctx.save_for_backward(x, xw, w)
        ctx.group, ctx.has_bias, ctx.x3 = group, b is not None, x3_planes
        ctx.mark_non_differentiable(part)
        ctx.set_materialize_grads(False)
        return y, part

    @staticmethod
    def backward(ctx, g, _gpart=None):
        x, xw, w = ctx.saved_tensors
        R, K = x.shape
        g = _req(g, torch.float32, "fan_linear_rows.grad")
        grp, ret = ctx.group, None
        if ctx.needs_input_grad[0]:
            wk = w if w.shape[1] == K else w[:, :K]
            with x3_scope(ctx.x3):
                if grp.gx is None:
                    grp.gx = ret = torch.empty(R, K, dtype=torch.float32, device=x.device)
                    gemm_own(g, wk, True, out=grp.gx)
                    # the buffer lives for THIS backward pass only: if a member's backward never runs in it (an unused head
                    # output, autograd.grad on a subset of the outputs) the next pass must not find it and add into it
                    torch.autograd.Variable._execution_engine.queue_callback(grp.end_of_pass)
                else:
                    gemm_own(g, wk, True, resid=grp.gx, out=grp.gx)
        gwt, gb = wgrad(x if xw is None else xw, g, colsum=True)                     # (Cin, Cout) = dW^T, column sums of g = db
        grp.seen += 1
        if grp.members and grp.seen >= grp.members:          # last member of this pass: a second backward over the same graph
            grp.end_of_pass()                                # (retain_graph, gradient checks) starts a fresh buffer
        return ret, None, None, gwt.t(), (gb if ctx.has_bias else None)


def fan_linear_rows_ok(x, xyz, weights):
    """shapes ``fan_linear_rows`` takes on the hand-written kernels (else the caller keeps one ``linear_rows`` per layer)"""
    if not GEMM_X3 or x.dtype != torch.float32 or not x.is_cuda or x.dim() != 2:
        return False
    R, K = x.shape
    if R < 1024 or x.stride(1) != 1 or x.data_ptr() % 16 or (x.stride(0) * 4) % 16 or x.stride(0) < (K + 3) // 4 * 4:
        return False
    for w in weights:
        if w.dim() != 2 or w.shape[0] % 128 or w.shape[1] not in (K, K + 3) or (w.shape[1] == K + 3 and xyz is None):
            return False
        if not _wgrad_ragged_ok(x, torch.empty(0, w.shape[0], device=x.device), x) and not (K % 64 == 0 and w.shape[0] % 64 == 0):
            return False
    return True


def fan_linear_rows(x, xyz, layers):
    """[F.linear(x or cat[x, xyz], W_i, b_i) for (W_i, b_i) in layers] for layers that share their input rows x (R, K) -- a
    weight with K + 3 columns reads cat[x, xyz], xyz (B, N, 3) with R = B N -- whose input gradients meet in the products'
    epilogues instead of in autograd's element-wise adds (``_FanMember``).  Each entry of the result is (y, part): part = the
    first pass of the train-mode BatchNorm behind the layer (``bn_relu(y, bn, partial=part)``), possibly empty."""
    R, K = x.shape
    # the members read x through ONE alias: autograd then delivers their (single, in-place accumulated) gradient to the alias node
    # only after every member has run, and adds a gradient x may receive from a consumer OUTSIDE the group after that -- the
    # buffer is complete before anything else touches it
    x = x.view_as(x)
    group, xw, outs = _FanGroup(), None, []
    group.members = len(layers)
    for w, b in layers:
        if w.shape[1] != K and xw is None:
            with torch.no_grad():                                 # (its gradient is routed by the member, not through the cat)
                xw = cat_rows_pitched([x, xyz.reshape(R, 3)])
        outs.append(_FanMember.apply(x, None if w.shape[1] == K else xw, group, w, b))
    return outs


class _CloudCatLinear(torch.autograd.Function):
    """F.linear(cat[fg[cloud] repeated over the cloud's rows, x, xyz], W, b) WITHOUT the concatenation (FaceRecon.py:113-117: the
    face head reads cat[f_global expanded to every point, the 256-wide block feature, the coordinates], 771 columns): the
    f_global columns are constant over a cloud, so their part of the product is a per-cloud bias
            y = cat[x, xyz] W[:, Cg:]^T + (fg W[:, :Cg]^T + b)[cloud]
    -- the trick HS_layer's conv2 already uses (gcn3d.py:186) -- a K = 259 product instead of K = 771 on a (R, 771) copy; in
    backward the f_global gradient and its weight block come from the per-cloud column sums of g (16 rows) instead of a sum over
    an expanded (B, N, 512) gradient, and the input gradient is a 512 -> 256 product instead of 512 -> 771."""

    @staticmethod
    def forward(ctx, fg, x, xyz, W, b):
        B, Cg = fg.shape
        R, Cx = x.shape
        N = R // B
        t = gemm_own(fg, W[:, :Cg], False)                        # (B, Cout): one small launch
        if b is not None:
            t = t + b
        xin = cat_rows_pitched([x, xyz.reshape(R, 3)])            # (R, Cx + 3) on a 16-byte pitch
        y = gemm_own(xin, W[:, Cg:], False, cloud_bias=t, rows_per_cloud=N)
        ctx.save_for_backward(fg, xin, W)
        ctx.dims, ctx.has_bias, ctx.x3 = (B, N, Cg, Cx), b is not None, x3_planes
        return y

    @staticmethod
    def backward(ctx, g):
        fg, xin, W = ctx.saved_tensors
        B, N, Cg, Cx = ctx.dims
        g = _req(g, torch.float32, "cloud_cat_linear.grad")
        Cout = g.shape[1]
        gt = colsum_rows(g.view(B, N, Cout))                      # (B, Cout) = the gradient of the per-cloud bias
        gW = torch.empty(Cout, Cg + Cx + 3, dtype=torch.float32, device=g.device)
        _tiny_tn(gt, fg, gW[:, :Cg])                              # the f_global block of dW
        with x3_scope(ctx.x3):
            g_fg = gemm_own(gt, W[:, :Cg], True) if ctx.needs_input_grad[0] else None
            gx = gemm_own(g, W[:, Cg:Cg + Cx], True) if ctx.needs_input_grad[1] else None
        sf = StepFolds.current                                    # (the result is re-laid out below: fold now, not at the step's end)
        held = sf.bare_wgrad if sf is not None else None
        if sf is not None:
            sf.bare_wgrad = False
        try:
            gwt = wgrad(xin, g)                                   # (Cx + 3, Cout) = the [x | xyz] block of dW^T
        finally:
            if sf is not None:
                sf.bare_wgrad = held
        gW[:, Cg:] = gwt.t()
        return g_fg, gx, None, gW, (gt.sum(0) if ctx.has_bias else None)


def cloud_cat_linear_ok(fg, x, xyz, W):
    B, Cg = fg.shape
    R, Cx = x.shape
    return (GEMM_X3 and fg.dtype == torch.float32 and x.dtype == torch.float32 and x.is_cuda and R % B == 0
            and R // B >= 128 and W.shape[1] == Cg + Cx + 3 and W.shape[0] % 128 == 0 and Cx % 4 == 0 and Cg % 4 == 0
)


def cloud_cat_linear(fg, x, xyz, W, b):
    """F.linear(cat[fg expanded over each cloud's rows (R, Cg), x (R, Cx), xyz (B, N, 3)], W (Cout, Cg + Cx + 3), b) as a K = Cx + 3
    product with a per-cloud bias (``_CloudCatLinear``)"""
    return _CloudCatLinear.apply(fg, x, xyz, W, b)


class _FaceSplit(torch.autograd.Function):
    """PoseNet9D.py:31-35 as one launch each way (``hsp_face_split_fwd / _bwd``)"""

    @staticmethod
    def forward(ctx, face):
        face = _req(face, torch.float32, "face_split.face")
        b, n, _ = face.shape
        nrm = torch.empty(b, n, 6, 3, dtype=torch.float32, device=face.device)
        dis = torch.empty(b, n, 6, dtype=torch.float32, device=face.device)
        conf = torch.empty(b, n, 6, dtype=torch.float32, device=face.device)
        _run("hsp_face_split_fwd", (_p(face), b * n, _p(nrm), _p(dis), _p(conf), _stream()), key=f"R{b * n}", abytes=4 * b * n * 60)
        ctx.save_for_backward(face)
        ctx.set_materialize_grads(False)
        return nrm, dis, conf

    @staticmethod
    def backward(ctx, gn, gd, gc):
        (face,) = ctx.saved_tensors
        if gn is None and gd is None and gc is None:
            return None
        b, n, _ = face.shape
        gs = [None if g is None else _req(g, torch.float32, "face_split.grad") for g in (gn, gd, gc)]
        gf = torch.empty_like(face)
        _run("hsp_face_split_bwd", (_p(face), _p(gs[0]), _p(gs[1]), _p(gs[2]), b * n, _p(gf), _stream()), key=f"R{b * n}",
             abytes=4 * b * n * 90)
        return gf


class _AxisConf(torch.autograd.Function):
    """PoseNet9D.py:40-46 as one launch each way (``hsp_axis_conf_fwd / _bwd``)"""

    @staticmethod
    def forward(ctx, h):
        h = _req(h, torch.float32, "axis_conf.h")
        B = h.shape[0]
        axis = torch.empty(B, 3, dtype=torch.float32, device=h.device)
        conf = torch.empty(B, dtype=torch.float32, device=h.device)
        _run("hsp_axis_conf_fwd", (_p(h), B, _p(axis), _p(conf), _stream()), key=f"B{B}")
        ctx.save_for_backward(h)
        ctx.set_materialize_grads(False)
        return axis, conf

    @staticmethod
    def backward(ctx, ga, gc):
        (h,) = ctx.saved_tensors
        if ga is None and gc is None:
            return None
        gs = [None if g is None else _req(g, torch.float32, "axis_conf.grad") for g in (ga, gc)]
        gh = torch.empty_like(h)
        _run("hsp_axis_conf_bwd", (_p(h), _p(gs[0]), _p(gs[1]), h.shape[0], _p(gh), _stream()), key=f"B{h.shape[0]}")
        return gh


def axis_conf(h):
    """(B, 4) rotation-head output -> (unit axis (B, 3) with the reference's 1e-6 guard, sigmoid confidence (B,)); PoseNet9D.py:40-46"""
    return _AxisConf.apply(h)


def face_split(face):
    """(B, N, 30) face-head output -> (unit normals (B, N, 6, 3), distances (B, N, 6), confidences (B, N, 6)); PoseNet9D.py:31-35"""
    return _FaceSplit.apply(face)


def cat_rows_pitched(parts):
    """torch.cat(parts, dim=-1) whose rows sit on a 16-byte pitch: the result is the (..., K) view of a (..., K rounded up to 4)
    buffer with zero pad columns, so the dense kernels that want aligned rows (csrc/gemm_x3.hip) take a K = 1289 / 1283 input
    (PoseTs.py:32 on cat[feat, xyz]; the face head on cat[f_global, h, xyz], FaceRecon.py:116) without a fallback"""
    K = sum(p.shape[-1] for p in parts)
    pad = (-K) % 4
    if pad == 0 or parts[0].dtype != torch.float32:
        return torch.cat(parts, dim=-1)
    z = torch.zeros(*parts[0].shape[:-1], pad, dtype=parts[0].dtype, device=parts[0].device)
    return torch.cat(list(parts) + [z], dim=-1)[..., :K]


def linear_rows(x2, weight, bias=None, bn_partials=False):
    """F.linear(x2, weight, bias) for (R, Cin) rows with a graph-replayable backward.  ``bn_partials``: returns (y, part) where
    ``part`` is the first pass of a train-mode BatchNorm over y left by the product's epilogue -- hand it to
    ``bn_relu(y, bn, partial=part)`` -- or an empty tensor when this shape / mode does not produce it."""
    if bn_partials:
        return _LinearRows.apply(x2, weight, bias, True)
    return _LinearRows.apply(x2, weight, bias)


# ------------------------------------------------------------------------------------------------
# fused train-mode BatchNorm1d + ReLU over point rows
# ------------------------------------------------------------------------------------------------

def _rows_pitch(t, C):
    """(tensor, row pitch in elements) when ``t`` (..., C) is a set of equally pitched rows a kernel can walk in place (contiguous,
    or an 8-byte aligned column block of a wider row-major tensor with an even pitch); else (a contiguous copy, C)"""
    ok = t.stride(-1) == 1 and t.data_ptr() % 8 == 0
    ld = t.stride(-2) if t.dim() >= 2 else C
    if ok and t.dim() >= 2:
        for d in range(t.dim() - 2):                          # leading dims must collapse onto the row pitch
            ok = ok and t.stride(d) == t.stride(d + 1) * t.shape[d + 1]
    if ok and ld >= C and ld % 2 == 0:
        return t, ld
    return t.contiguous(), C


class _BNRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, num_batches, eps, momentum, relu, out_dtype=None, fork=False,
                partial=None):
        x = _reqf(x, "bn_relu.x")
        C = x.shape[-1]
        R = x.numel() // C
        # "mixed": fp32 rows in (a pre-BatchNorm tensor keeps its mantissa: |mean| >> std per channel is common), bf16 out
        mixed = out_dtype == torch.bfloat16 and x.dtype == torch.float32
        y = torch.empty_like(x, dtype=torch.bfloat16) if mixed else torch.empty_like(x)
        ctx.mixed = mixed
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        L = lib()
        wsb = L.hsp_bn_workspace_bytes(R, C)
        ws = _ws(wsb, x.device)
        if partial is not None and partial.numel() and not mixed and x.dtype == torch.float32 and running_mean is not None:
            # the first pass (row 0 of ``partial``: the shift; then the shifted column sums per row tile) was left by the producer's
            # epilogue (hsp_gemm_x3_bn_f32): fold + apply only
            _run("hsp_bn_relu_fwd_partials", (_p(x), R, C, _p(weight), _p(bias), float(eps), float(momentum), 1 if relu else 0, _p(y),
                                              _p(mean), _p(invstd), _p(running_mean), _p(running_var), _p(num_batches),
                                              _p(partial[1:]), (partial.shape[0] - 1) // 2, _p(partial[0]), _stream()),
                 key=f"R{R}C{C}", abytes=8 * R * C)
        else:
            _run("hsp_bn_relu_fwd" + ("_mixed" if mixed else _sfx(x)), (_p(x), R, C, _p(weight), _p(bias), float(eps), float(momentum), 1 if relu else 0,
                                               _p(y), _p(mean), _p(invstd), _p(running_mean), _p(running_var), _p(num_batches),
                                               _p(ws), wsb, _stream()),
                 key=f"R{R}C{C}", abytes=(_es(x) + _es(y)) * R * C)
        ctx.save_for_backward(x, weight, bias, mean, invstd)
        ctx.relu = relu
        ctx.fork = fork
        ctx.mark_non_differentiable(*[t for t in (running_mean, running_var, num_batches) if t is not None])
        if fork:
            # TWO outputs over the same rows, one per consumer: autograd then hands their gradients to backward separately and the
            # BatchNorm kernels add them as they read (no element-wise add kernel, no copy of a strided column block)
            ctx.set_materialize_grads(False)
            return y, y.view_as(y)
        return y

    @staticmethod
    def backward(ctx, *dys):
        x, weight, bias, mean, invstd = ctx.saved_tensors
        C = x.shape[-1]
        R = x.numel() // C
        dys = [d for d in dys if d is not None]
        if not dys:
            return (None,) * 12
        L = lib()
        wsb = L.hsp_bn_workspace_bytes(R, C)
        ws = _ws(wsb, x.device)
        dg = torch.empty_like(weight)
        db = torch.empty_like(bias)
        if ctx.fork and not ctx.mixed and x.dtype == torch.float32 and all(d.is_cuda and d.dtype == torch.float32 for d in dys):
            (d0, ld0) = _rows_pitch(dys[0], C)
            (d1, ld1) = _rows_pitch(dys[1], C) if len(dys) > 1 else (None, 0)
            dx = torch.empty_like(x)
            _run("hsp_bn_relu_bwd2", (_p(x), _p(d0), ld0, _p(d1), ld1, R, C, _p(weight), _p(bias), _p(mean), _p(invstd),
                                      1 if ctx.relu else 0, _p(dx), _p(dg), _p(db), _p(ws), wsb, _stream()),
                 key=f"R{R}C{C}", abytes=4 * (2 + len(dys)) * R * C)
            return dx, dg, db, None, None, None, None, None, None, None, None, None
        dy = dys[0] if len(dys) == 1 else dys[0] + dys[1]
        dy = _reqf(dy, "bn_relu.grad", like=None if ctx.mixed else x)
        dx = torch.empty_like(dy)
        _run("hsp_bn_relu_bwd" + ("_mixed" if ctx.mixed else _sfx(x)), (_p(x), _p(dy), R, C, _p(weight), _p(bias), _p(mean), _p(invstd),
                                           1 if ctx.relu else 0, _p(dx), _p(dg), _p(db), _p(ws), wsb, _stream()),
             key=f"R{R}C{C}", abytes=(_es(x) + 2 * _es(dy)) * R * C)
        return dx, dg, db, None, None, None, None, None, None, None, None, None


def _eval_invstd(bn):
    """1 / sqrt(running_var + eps) exactly as the reference's eval-mode BatchNorm gets it: from the HOST's ATen
    (batch_norm_cpu_transform_input_template evaluates ``1 / at::sqrt(running_var + eps)``; at::sqrt is MKL VML's vsSqrt, which
    is within an ulp but not correctly rounded, so no device formula reproduces it).  C floats, cached on the module until the
    running variance changes; None inside a stream capture that finds no cached value (the kernel then uses the correctly
    rounded value)."""
    rv = bn.running_var
    key = (rv.data_ptr(), rv._version, float(bn.eps), rv.device)
    hit = getattr(bn, "_hsp_invstd", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    if torch.cuda.is_current_stream_capturing():
        return None
    inv = (1 / torch.sqrt(rv.detach().cpu() + bn.eps)).to(rv.device)
    bn._hsp_invstd = (key, inv)
    return inv


class _BNEval(torch.autograd.Function):
    """eval-mode BatchNorm (+ ReLU) on point rows, hsp_bn_eval_f32; the backward (rare: an eval-mode network being
    differentiated) is the textbook affine one in torch ops"""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, invstd, eps, relu):
        C = x.shape[-1]
        y = torch.empty_like(x)
        _run("hsp_bn_eval_f32", (_p(x), x.numel() // C, C, _p(running_mean), _p(running_var), _p(invstd), _p(weight), _p(bias),
                                 float(eps), 1 if relu else 0, _p(y), _stream()), key=f"R{x.numel() // C}C{C}", abytes=8 * x.numel())
        ctx.save_for_backward(x, weight, running_mean, invstd if invstd is not None else torch.rsqrt(running_var + eps), y)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, mean, invstd, y = ctx.saved_tensors
        if ctx.relu:
            dy = dy * (y > 0)
        C = x.shape[-1]
        xh = ((x - mean) * invstd).reshape(-1, C)
        d2 = dy.reshape(-1, C)
        return dy * (invstd * weight), (d2 * xh).sum(0), d2.sum(0), None, None, None, None, None


def bn_relu(x, bn, relu=True, out_dtype=None, fork=False, partial=None):
    """relu(bn(x)) for point rows x (..., C) with an nn.BatchNorm1d module ``bn`` (its parameters, running
    statistics and train/eval state are honoured exactly like calling the module on the (R,C) view, which
    is what the reference's transpose->BatchNorm1d->transpose computes, FaceRecon.py:90-95).  ``fork``: return the result
    TWICE (two tensors over the same rows) for a tensor with two consumers -- their gradients then reach the BatchNorm backward
    separately and are added inside its kernels."""
    C = x.shape[-1]
    fused = (bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None and x.is_cuda
             and x.dtype in _FEAT_DTYPES and C % 4 == 0 and 256 % (C // 4) == 0)
    want_bf16 = x.dtype == torch.bfloat16 or out_dtype == torch.bfloat16
    if not fused and want_bf16:                      # eval mode, bf16 rows out: running statistics, fused affine + ReLU
        if bn.training or not bn.affine or C % 4 or 256 % (C // 4):
            raise HspError("bn_relu: this BatchNorm configuration is not built for bf16 rows")
        xc = _reqf(x, "bn_relu.x")
        y = torch.empty_like(xc, dtype=torch.bfloat16)
        invstd = torch.rsqrt(bn.running_var + bn.eps)
        _run("hsp_bn_relu_apply" + ("_bf16" if xc.dtype == torch.bfloat16 else "_mixed"),
             (_p(xc), xc.numel() // C, C, _p(bn.running_mean), _p(invstd), _p(bn.weight), _p(bn.bias), 1 if relu else 0, _p(y),
              _stream()), key=f"R{xc.numel() // C}C{C}", abytes=(_es(xc) + 2) * xc.numel())
        return (y, y) if fork else y
    if (not fused and not bn.training and bn.affine and bn.track_running_stats and x.is_cuda and x.dtype == torch.float32
            and C % 4 == 0 and x.is_contiguous()):
        # eval mode: ((x - m) * invstd) * w + b in ATen's own operation order (hsp_bn_eval_f32), relu fused
        if _timer is None and not torch.is_grad_enabled() and _ext_ok():
            from ._ext import ext
            y = ext().bn_eval(x, bn.running_mean, bn.running_var, _eval_invstd(bn), bn.weight, bn.bias, bn.eps, relu)
            return (y, y.view_as(y)) if fork else y
        y = _BNEval.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, _eval_invstd(bn), bn.eps, relu)
        return (y, y.view_as(y)) if fork else y
    if not fused:                                   # eval mode / exotic configurations: not on the training hot path
        y = bn(x.reshape(-1, C)).view_as(x)
        y = torch.relu_(y) if relu else y
        return (y, y) if fork else y
    return _BNRelu.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.eps,
                         bn.momentum, relu, out_dtype, fork, partial)


# ------------------------------------------------------------------------------------------------
# row gather
# ------------------------------------------------------------------------------------------------

class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, idx):
        feat = _req(feat, torch.float32, "gather_rows.feat")
        idx = _req(idx, torch.int32, "gather_rows.idx")
        B, Nsrc, C = feat.shape
        shared = 1 if idx.dim() == 1 else 0
        Nq = idx.shape[-1]
        out = torch.empty(B, Nq, C, dtype=torch.float32, device=feat.device)
        _run("hsp_gather_rows_fwd", (_p(feat), _p(idx), shared, B, Nsrc, Nq, C, _p(out), C, _stream()),
             key=f"B{B}Ns{Nsrc}Nq{Nq}C{C}", abytes=B * (4 * Nsrc * C + Nq * (4 + 4 * C)))
        ctx.save_for_backward(idx)
        ctx.dims = (B, Nsrc, Nq, C, shared)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        B, Nsrc, Nq, C, shared = ctx.dims
        if not (g.is_cuda and g.dtype == torch.float32):
            raise HspError("gather_rows.grad: expected a float32 GPU tensor")
        # a column slice of a wider row-major tensor (the grad of torch.cat) is consumed in place
        if g.stride(2) == 1 and g.stride(0) == Nq * g.stride(1) and g.stride(1) >= C:
            gstride = g.stride(1)
        else:
            g = g.contiguous()
            gstride = C
        gfeat = torch.empty(B, Nsrc, C, dtype=torch.float32, device=g.device)
        _run("hsp_gather_rows_bwd", (_p(g), gstride, _p(idx), shared, B, Nsrc, Nq, C, _p(gfeat), _stream()),
             key=f"B{B}Ns{Nsrc}Nq{Nq}C{C}", abytes=B * (4 * Nsrc * C + Nq * (4 + 4 * C)))
        return gfeat, None


def gather_rows(feat, idx):
    """feat (B,Nsrc,C), idx int32 (B,Nq) or shared (Nq,) -> (B,Nq,C)."""
    return _GatherRows.apply(feat, idx)


# ------------------------------------------------------------------------------------------------
# feat assembly: nearest up-sampling + one-hot + concat in one kernel (FaceRecon.py:100-107)
# ------------------------------------------------------------------------------------------------

ONE_HOT_WIDTH = 6          # categories of a kind-3 (one-hot) segment; FaceRecon sets it from FLAGS.obj_c

# feat row pitch: columns padded to a multiple of this many elements (8 = 16 bytes of bf16, 32 of fp32); 1 = no padding
FEAT_PITCH_ALIGN = 8


class _AssembleFeat(torch.autograd.Function):
    """feat (B,N,W) = cat[ direct..., gathered(src, nearest idx)..., per-cloud... ] along channels.
    segs: list of (tensor, idx or None, kind) with kind 0 direct (B,N,w), 1 gathered rows of (B,Ns,w) by an
    int32 (B,N) index, 2 per-cloud (B,w) broadcast over the points, 3 a (B,) float id expanded to ``ONE_HOT_WIDTH`` one-hot
    columns (the reference's zeros + scatter_, FaceRecon.py:80-85, without its three launches)."""

    @staticmethod
    def forward(ctx, kinds, idxs, *tensors):
        B, N, dt = None, None, None
        for t, kd in zip(tensors, kinds):
            if kd == 0:
                B, N, dt = t.shape[0], t.shape[1], t.dtype
        # point-row segments share the feature dtype (fp32 or bf16); per-cloud rows (kind 2: the one-hot columns) are fp32
        tensors = [(_req(t, torch.float32, "assemble_feat.src") if kd >= 2 or dt == torch.float32 else
                    _req(t, dt, "assemble_feat.src")) for t, kd in zip(tensors, kinds)]
        idxs = [(_req(i, torch.int32, "assemble_feat.idx") if i is not None else None) for i in idxs]
        n = len(tensors)
        widths = [(t.shape[-1] if kd != 3 else ONE_HOT_WIDTH) for t, kd in zip(tensors, kinds)]
        for t, kd in zip(tensors, kinds):
            if kd == 3 and t.numel() != B:
                raise HspError("assemble_feat: a kind-3 segment is one float id per cloud")
        W = sum(widths)
        # rows padded to a multiple of 16 bytes (1286 -> 1288 columns): the kernel then moves 16 bytes per access and the
        # heads' K = 1286 products read aligned rows; the result is the (B,N,W) view of the padded buffer
        P = (W + FEAT_PITCH_ALIGN - 1) // FEAT_PITCH_ALIGN * FEAT_PITCH_ALIGN if FEAT_PITCH_ALIGN > 1 else W
        full = torch.empty(B, N, P, dtype=dt, device=tensors[0].device)
        out = full[:, :, :W] if P != W else full
        # (the kernel zeroes the pad columns: never read as data, but a consumer that loads whole 16-byte chunks touches them)
        src = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tensors])
        ix = (ctypes.c_void_p * n)(*[(i.data_ptr() if i is not None else 0) for i in idxs])
        wd = (ctypes.c_int * n)(*widths)
        kd = (ctypes.c_int * n)(*kinds)
        ns = (ctypes.c_int * n)(*[(t.shape[1] if k_ == 1 else 0) for t, k_ in zip(tensors, kinds)])
        es = 2 if dt == torch.bfloat16 else 4
        if dt == torch.bfloat16:
            _run("hsp_concat_rows_bf16", (n, ctypes.cast(src, _vp), ctypes.cast(ix, _vp), ctypes.cast(wd, _vp),
                                          ctypes.cast(kd, _vp), ctypes.cast(ns, _vp), B, N, _p(full), P, _stream()),
                 key=f"B{B}N{N}W{W}", abytes=2 * es * B * N * W)
        else:
            _run("hsp_concat_rows_pitched", (n, ctypes.cast(src, _vp), ctypes.cast(ix, _vp), ctypes.cast(wd, _vp),
                                             ctypes.cast(kd, _vp), ctypes.cast(ns, _vp), B, N, _p(full), P, _stream()),
                 key=f"B{B}N{N}W{W}", abytes=2 * es * B * N * W)
        ctx.kinds, ctx.widths = kinds, widths
        ctx.nsrc = [(t.shape[1] if t.dim() > 1 else 0) for t in tensors]
        ctx.save_for_backward(*[i for i in idxs if i is not None])
        ctx.has_idx = [i is not None for i in idxs]
        return out

    @staticmethod
    def backward(ctx, g):
        if not g.is_contiguous():
            g = g.contiguous()
        B, N, W = g.shape
        saved = list(ctx.saved_tensors)
        grads, col = [], 0
        for s_, (kd, w) in enumerate(zip(ctx.kinds, ctx.widths)):
            gs = g[:, :, col:col + w]
            if not ctx.needs_input_grad[2 + s_]:
                grads.append(None)
                if ctx.has_idx[s_]:
                    saved.pop(0)
            elif kd == 0:
                grads.append(gs)                                   # strided view: consumers take it as is / copy
            elif kd == 1:
                idx = saved.pop(0)
                Ns = ctx.nsrc[s_]
                gfeat = torch.empty(B, Ns, w, dtype=g.dtype, device=g.device)
                es = _es(g)
                if w % 2 == 0 and W % 2 == 0 and (w // 2 >= 256 or 256 % (w // 2) == 0) and gs.data_ptr() % (2 * es) == 0:
                    # gather form over the reverse map (memoised on idx: fm_2 and fm_3 share one); deterministic
                    off, edge = rev_index(idx, 1, Ns)
                    _run("hsp_gather_rows_bwd_csr" + _sfx(g), (_p(gs), W, _p(off), _p(edge), B, Ns, N, w, _p(gfeat), _stream()),
                         key=f"B{B}Ns{Ns}Nq{N}C{w}", abytes=B * (es * Ns * w + N * (4 + es * w)))
                elif g.dtype == torch.bfloat16:
                    raise HspError("assemble_feat backward: bf16 segments must be even-width and 4-byte aligned")
                else:
                    _run("hsp_gather_rows_bwd", (_p(gs), W, _p(idx), 0, B, Ns, N, w, _p(gfeat), _stream()),
                         key=f"B{B}Ns{Ns}Nq{N}C{w}", abytes=B * (4 * Ns * w + N * (4 + 4 * w)))
                grads.append(gfeat)
            elif kd == 3:
                grads.append(None)                                 # ids carry no gradient
            else:
                grads.append(gs.sum(dim=1))
            col += w
        return (None, None, *grads)


def assemble_feat(segments):
    """segments: list of (tensor, idx_or_None, kind) -> (B,N,sum widths); see _AssembleFeat.  NOTE the result is the (B,N,W)
    column slice of a buffer whose rows are padded to a multiple of 16 bytes (``FEAT_PITCH_ALIGN``; 1286 -> 1288 columns, pad
    columns zeroed): row-strided, not contiguous.  The heads' ``linear_rows`` / ``gemm_rows`` take the row stride as is; a
    consumer that calls ``.contiguous()`` pays a copy of the whole tensor."""
    tensors = [s_[0] for s_ in segments]
    idxs = [s_[1] for s_ in segments]
    kinds = [s_[2] for s_ in segments]
    return _AssembleFeat.apply(kinds, idxs, *tensors)


# ------------------------------------------------------------------------------------------------
# Chamfer / FPS
# ------------------------------------------------------------------------------------------------

class _Chamfer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        xyz1 = _req(xyz1, torch.float32, "chamfer.xyz1")
        xyz2 = _req(xyz2, torch.float32, "chamfer.xyz2")
        B, n, _ = xyz1.shape
        m = xyz2.shape[1]
        d1 = torch.empty(B, n, dtype=torch.float32, device=xyz1.device)
        d2 = torch.empty(B, m, dtype=torch.float32, device=xyz1.device)
        i1 = torch.empty(B, n, dtype=torch.int32, device=xyz1.device)
        i2 = torch.empty(B, m, dtype=torch.int32, device=xyz1.device)
        _run("hsp_chamfer_fwd", (_p(xyz1), _p(xyz2), B, n, m, _p(d1), _p(d2), _p(i1), _p(i2), _stream()),
             key=f"B{B}n{n}m{m}", abytes=B * (n + m) * 20)
        ctx.save_for_backward(xyz1, xyz2, i1, i2)
        ctx.mark_non_differentiable(i1, i2)
        return d1, d2, i1, i2

    @staticmethod
    def backward(ctx, g1, g2, _gi1, _gi2):
        xyz1, xyz2, i1, i2 = ctx.saved_tensors
        B, n, _ = xyz1.shape
        m = xyz2.shape[1]
        g1 = _req(g1 if g1 is not None else torch.zeros(B, n, device=xyz1.device), torch.float32, "chamfer.g1")
        g2 = _req(g2 if g2 is not None else torch.zeros(B, m, device=xyz1.device), torch.float32, "chamfer.g2")
        gx1 = torch.empty_like(xyz1)
        gx2 = torch.empty_like(xyz2)
        _run("hsp_chamfer_bwd", (_p(xyz1), _p(xyz2), _p(i1), _p(i2), _p(g1), _p(g2), B, n, m, _p(gx1), _p(gx2),
                                 _stream()), key=f"B{B}n{n}m{m}", abytes=B * (n + m) * 32)
        return gx1, gx2


def chamfer(xyz1, xyz2):
    """(dist1, dist2, idx1, idx2): squared NN distances both ways + int32 arg-mins."""
    return _Chamfer.apply(xyz1, xyz2)


def fps(xyz, n_samples):
    """int32 (B,n_samples) farthest-point-sampling indices per cloud (first pick = index 0); float32 or float64 clouds,
    computed in the cloud's own dtype like the numpy helper does (tools/eval_utils.py:73-84,107-119)."""
    f64 = isinstance(xyz, torch.Tensor) and xyz.dtype == torch.float64
    xyz = _req(xyz.detach(), torch.float64 if f64 else torch.float32, "fps.xyz")
    B, N, _ = xyz.shape
    sel = torch.empty(B, n_samples, dtype=torch.int32, device=xyz.device)
    L = lib()
    wsb = L.hsp_fps_workspace_bytes(B, N) * (2 if f64 else 1)
    ws = _ws(wsb, xyz.device)
    _run("hsp_fps_f64" if f64 else "hsp_fps_f32", (_p(xyz), B, N, n_samples, _p(sel), _p(ws), wsb, _stream()),
         key=f"B{B}N{N}n{n_samples}", abytes=B * ((24 if f64 else 12) * N + 4 * n_samples))
    return sel


# ------------------------------------------------------------------------------------------------
# inference front / back end: depth -> cloud, axes -> pose matrix (no gradient)
# ------------------------------------------------------------------------------------------------

def pc_compact(mask, depth):
    """mask, depth (B,HW) fp32 -> (pix (B,HW) int32, count (B) int32): ids of the pixels with
    mask * (depth > 0) > 0, row-major (pc_sample.py:38-39, :52-54); entries past count[b] are undefined."""
    mask = _req(mask.detach(), torch.float32, "pc_compact.mask")
    depth = _req(depth.detach(), torch.float32, "pc_compact.depth")
    if mask.dim() != 2 or mask.shape != depth.shape:
        raise HspError("pc_compact: expects mask and depth of the same (B,HW) shape")
    B, HW = mask.shape
    pix = torch.empty(B, HW, dtype=torch.int32, device=mask.device)
    count = torch.empty(B, dtype=torch.int32, device=mask.device)
    L = lib()
    wsb = L.hsp_pc_compact_workspace_bytes(B, HW)
    ws = _ws(wsb, mask.device)
    _run("hsp_pc_compact", (_p(mask), _p(depth), B, HW, _p(pix), _p(count), _p(ws), wsb, _stream()),
         key=f"B{B}HW{HW}", abytes=B * HW * 12)
    return pix, count


def pc_gather(depth, coor2d, camK, pix, choose):
    """back-project the chosen valid pixels: (B,S,3) metres (pc_sample.py:41-50, :66, :77)."""
    depth = _req(depth.detach(), torch.float32, "pc_gather.depth")
    coor2d = _req(coor2d.detach(), torch.float32, "pc_gather.coor2d")
    camK = _req(camK.detach(), torch.float32, "pc_gather.camK")
    pix = _req(pix, torch.int32, "pc_gather.pix")
    choose = _req(choose, torch.int32, "pc_gather.choose")
    B, HW = depth.shape
    if coor2d.shape != (B, 2, HW) or camK.shape != (B, 3, 3) or pix.shape != (B, HW) or choose.shape[0] != B:
        raise HspError("pc_gather: expects depth (B,HW), coor2d (B,2,HW), camK (B,3,3), pix (B,HW), choose (B,S)")
    S = choose.shape[1]
    pc = torch.empty(B, S, 3, dtype=torch.float32, device=depth.device)
    _run("hsp_pc_gather", (_p(depth), _p(coor2d), _p(camK), _p(pix), _p(choose), B, HW, S, _p(pc), _stream()),
         key=f"B{B}S{S}", abytes=B * S * 32)
    return pc


def generate_rt(p_green, p_red, f_green, f_red, T, sym):
    """(B,4,4) pose matrices; semantics of geom_utils.generate_RT(mode='vec')."""
    pg = _req(p_green.detach(), torch.float32, "generate_rt.p_green")
    pr = _req(p_red.detach(), torch.float32, "generate_rt.p_red")
    fg = _req(f_green.detach(), torch.float32, "generate_rt.f_green")
    fr = _req(f_red.detach(), torch.float32, "generate_rt.f_red")
    T = _req(T.detach(), torch.float32, "generate_rt.T")
    sym = _req(sym.detach().float(), torch.float32, "generate_rt.sym")
    B = pg.shape[0]
    if pg.shape != (B, 3) or pr.shape != (B, 3) or T.shape != (B, 3) or fg.numel() != B or fr.numel() != B \
            or sym.dim() != 2 or sym.shape[0] != B:
        raise HspError("generate_rt: expects p_green/p_red/T (B,3), f_green/f_red (B), sym (B,>=1)")
    out = torch.empty(B, 4, 4, dtype=torch.float32, device=pg.device)
    _run("hsp_generate_rt", (_p(pg), _p(pr), _p(fg), _p(fr), _p(T), _p(sym), sym.shape[1], B, _p(out), _stream()),
         key=f"B{B}", abytes=B * (11 * 4 + 64))
    return out


def depth_to_pcl(depth, xymap, camK64, pix, choose):
    """dataset-side back-projection (load_data.py:322-333 then / 1000.0): float64 arithmetic with a float64
    K (B,3,3), fp32 (B,S,3) result."""
    depth = _req(depth.detach(), torch.float32, "depth_to_pcl.depth")
    xymap = _req(xymap.detach(), torch.float32, "depth_to_pcl.xymap")
    camK64 = _req(camK64.detach(), torch.float64, "depth_to_pcl.camK")
    pix = _req(pix, torch.int32, "depth_to_pcl.pix")
    choose = _req(choose, torch.int32, "depth_to_pcl.choose")
    B, HW = depth.shape
    if xymap.shape != (B, 2, HW) or camK64.numel() != B * 9 or pix.shape != (B, HW) or choose.shape[0] != B:
        raise HspError("depth_to_pcl: expects depth (B,HW), xymap (B,2,HW), camK (B,3,3) f64, pix (B,HW), choose (B,S)")
    S = choose.shape[1]
    pc = torch.empty(B, S, 3, dtype=torch.float32, device=depth.device)
    _run("hsp_depth_to_pcl", (_p(depth), _p(xymap), _p(camK64), _p(pix), _p(choose), B, HW, S, _p(pc), _stream()),
         key=f"B{B}S{S}", abytes=B * S * 32)
    return pc
