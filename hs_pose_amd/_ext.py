"""Loader of the thin C++ PyTorch binding over libhsp.so (csrc/hsp_torch.cpp -> hs_pose_amd/_hsp_torch.so, built in-tree by
``make -C hs_pose_amd/csrc`` / ``__graft_entry__.build()``).

The binding carries the reference's own extension surface (tools/pyTorchChamferDistance/chamfer_distance.cpp:180-185:
``forward`` / ``forward_cuda`` / ``backward`` / ``backward_cuda``) and the eval-mode HS layers as one call each; the ctypes
binding (``_lib.py``) stays the training path's and the ABI tests'.  ``ext()`` raises when the module is missing; the inference
forms in ``ops.py`` ask ``ops._ext_ok()`` first and issue the same libhsp.so launches through ctypes otherwise (still the HIP
kernels: there is no CPU path anywhere); ``chamfer.py`` has no ctypes twin and lets the error through."""
import importlib
import os

_mod = None


class HspExtError(RuntimeError):
    pass


def ext():
    """the ``_hsp_torch`` module (imported once)"""
    global _mod
    if _mod is None:
        import torch  # noqa: F401  (libtorch must be loaded before the extension resolves its symbols)
        from ._lib import lib
        lib()                                          # libhsp.so first: the extension's NEEDED entry then binds to this instance
        try:
            _mod = importlib.import_module("hs_pose_amd._hsp_torch")
        except ImportError as e:
            here = os.path.dirname(os.path.abspath(__file__))
            raise HspExtError(f"hs_pose_amd/_hsp_torch.so is missing or does not load ({e}); build it with "
                              f"`make -C {os.path.join(here, 'csrc')}` or __graft_entry__.build()") from e
    return _mod


def available():
    """True when the built module is present (the product paths that need it call ``ext()`` and raise otherwise)"""
    here = os.path.dirname(os.path.abspath(__file__))
    return os.path.exists(os.path.join(here, "_hsp_torch.so"))
