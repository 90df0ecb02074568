import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def tiled_batch(ref, bases, seed, n_pts=1028):
    """the inputs of oracle/gen_golden_tiled.py, rebuilt (oracle/ref_cpu.py::tiled_batch) so the fixtures carry outputs only"""
    return ref.tiled_batch(bases, seed, n_pts)


@pytest.fixture(scope="session")
def oracle_lib():
    """oracle/libhsp_oracle.so (C index oracle), built on demand with gcc."""
    so = os.path.join(ROOT, "oracle", "libhsp_oracle.so")
    src = os.path.join(ROOT, "oracle", "hsp_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    return ctypes.CDLL(so)


def _P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class OracleC:
    """numpy-facing wrapper of the C oracle."""

    def __init__(self, lib):
        self.lib = lib

    def knn(self, x, k, drop_first=1, with_dist=False):
        xn = np.ascontiguousarray(x, dtype=np.float32)
        B, N, C = xn.shape
        out = np.empty((B, N, k), np.int32)
        ds = np.empty((B, N, k), np.float32) if with_dist else None
        rc = self.lib.hsp_oracle_knn(_P(xn), B, N, C, k, drop_first, _P(out), _P(ds) if with_dist else None)
        assert rc == 0
        return (out, ds) if with_dist else out

    def knn_topk(self, x, k, drop_first=1):
        """get_neighbor_index with torch.topk's own order among exactly equal distances (what the reference returns)"""
        xn = np.ascontiguousarray(x, dtype=np.float32)
        B, N, C = xn.shape
        out = np.empty((B, N, k), np.int32)
        assert self.lib.hsp_oracle_knn_topk(_P(xn), B, N, C, k, drop_first, _P(out)) == 0
        return out

    def topk_smallest(self, d, m):
        """indices torch.topk(d, m, largest=False) returns on the CPU, ties included (libstdc++'s order, restated in C)"""
        dn = np.ascontiguousarray(d, dtype=np.float32)
        R, N = dn.shape
        out = np.empty((R, m), np.int32)
        self.lib.hsp_oracle_topk_smallest(_P(dn), R, N, m, _P(out))
        return out

    def nn1(self, t, s):
        tn = np.ascontiguousarray(t, dtype=np.float32)
        sn = np.ascontiguousarray(s, dtype=np.float32)
        out = np.empty(tn.shape[:2], np.int32)
        self.lib.hsp_oracle_nn1(_P(tn), tn.shape[1], _P(sn), sn.shape[1], tn.shape[0], tn.shape[2], _P(out))
        return out

    def quad(self, x):
        xn = np.ascontiguousarray(x, dtype=np.float32)
        rows, C = xn.shape
        out = np.empty(rows, np.float32)
        self.lib.hsp_oracle_quad(_P(xn), ctypes.c_int64(rows), C, _P(out))
        return out

    def chamfer_fwd(self, x1, x2):
        a = np.ascontiguousarray(x1, dtype=np.float32)
        b = np.ascontiguousarray(x2, dtype=np.float32)
        B, n, _ = a.shape
        m = b.shape[1]
        d1, d2 = np.empty((B, n), np.float32), np.empty((B, m), np.float32)
        i1, i2 = np.empty((B, n), np.int32), np.empty((B, m), np.int32)
        self.lib.hsp_oracle_chamfer_fwd(_P(a), _P(b), B, n, m, _P(d1), _P(d2), _P(i1), _P(i2))
        return d1, d2, i1, i2

    def chamfer_bwd(self, x1, x2, i1, i2, g1, g2):
        a = np.ascontiguousarray(x1, dtype=np.float32)
        b = np.ascontiguousarray(x2, dtype=np.float32)
        B, n, _ = a.shape
        m = b.shape[1]
        gx1, gx2 = np.empty_like(a), np.empty_like(b)
        self.lib.hsp_oracle_chamfer_bwd(_P(a), _P(b), _P(np.ascontiguousarray(i1, np.int32)), _P(np.ascontiguousarray(i2, np.int32)),
                                        _P(np.ascontiguousarray(g1, np.float32)), _P(np.ascontiguousarray(g2, np.float32)),
                                        B, n, m, _P(gx1), _P(gx2))
        return gx1, gx2

    def fps_f32(self, pts, n_samples):
        p = np.ascontiguousarray(pts, dtype=np.float32)
        B, N, _ = p.shape
        out = np.empty((B, n_samples), np.int32)
        self.lib.hsp_oracle_fps_f32(_P(p), B, N, n_samples, _P(out))
        return out

    def fps_f64(self, pts, n_samples):
        p = np.ascontiguousarray(pts, dtype=np.float64)
        B, N, _ = p.shape
        out = np.empty((B, n_samples), np.int32)
        self.lib.hsp_oracle_fps_f64(_P(p), B, N, n_samples, _P(out))
        return out


@pytest.fixture(scope="session")
def oc(oracle_lib):
    return OracleC(oracle_lib)


@pytest.fixture(scope="session")
def ref():
    import ref_cpu
    return ref_cpu


@pytest.fixture(scope="session")
def state_keys():
    with open(os.path.join(GOLD, "state_keys.json")) as f:
        return json.load(f)


@pytest.fixture()
def flags():
    from hs_pose_amd.config import FLAGS
    FLAGS.reset()
    yield FLAGS
    FLAGS.reset()


@pytest.fixture(scope="session")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from hs_pose_amd._lib import lib
    lib()   # hard error (not a skip) if the HIP extension is missing on a GPU box
    return torch.device("cuda:0")


@pytest.fixture(params=["library", "own"])
def gemm_mode(request, monkeypatch):
    """the fp32 dense products of the layer / head nodes on the BLAS library through torch, or on the hand-written kernels only
    (csrc/gemm_wave.hip, gemm_rows.hip, gemm.hip): the stack goldens hold for both"""
    if request.param == "library":
        sys.path.insert(0, ROOT) if ROOT not in sys.path else None
        from tools import library_gemm
        library_gemm.enable(monkeypatch)          # the comparison shim: BLAS-library composites, own-only fusions off
    return request.param
