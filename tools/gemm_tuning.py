"""COMPARISON TOOLING (not on the product path: the shipped step runs no BLAS-library GEMM, hs_pose_amd/ops.py HSP_GEMM=own).
Library-GEMM solution selection for the `HSP_GEMM=library` baseline figure of bench.py and tools/.

The GEMMs that are not hand-written (everything except the split-K weight-gradient kernel in
csrc/gemm.hip) go to hipBLASLt / rocBLAS through torch.  Their default heuristics pick poor tiles for
the tall-skinny fp32 shapes of this workload (measured 37-50 TF/s); PyTorch's TunableOp times the
available solutions per shape and keeps the best (85-108 TF/s on the large ones).  ``enable()`` turns
it on with the MI355X table kept in tools/tuning/ and lets it tune any shape that is missing
during the first (eager, un-captured) iterations.
"""
import os
import shutil
import tempfile

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SHIPPED = os.path.join(_HERE, "tuning", "tunableop_gfx950.csv")
_enabled = False


def enable(tune_missing=True, max_tuning_ms=30):
    """idempotent; returns the path of the per-process results file TunableOp reads/writes."""
    global _enabled
    if _enabled or not torch.cuda.is_available():
        return None
    tun = torch.cuda.tunable
    # HSP_TUNABLEOP_OUT=<path>: keep the merged table there (used to regenerate the shipped one)
    work = os.environ.get("HSP_TUNABLEOP_OUT") or os.path.join(tempfile.gettempdir(), f"hsp_tunableop_{os.getpid()}.csv")
    if os.path.exists(SHIPPED):
        shutil.copyfile(SHIPPED, work)          # never write into the source tree
    tun.enable(True)
    tun.set_filename(work, insert_device_ordinal=False)
    tun.tuning_enable(bool(tune_missing))
    tun.set_max_tuning_duration(max_tuning_ms)
    for name, arg in (("write_file_on_exit", False), ("record_untuned_enable", False)):
        fn = getattr(tun, name, None)            # present in some torch versions only
        if fn is not None:
            try:
                fn(arg)
            except Exception:
                pass
    try:
        tun.read_file(work)
    except Exception:
        pass
    _enabled = True
    return work


def save(path=None):
    """write the merged table (shipped entries + what this process tuned) -- how tools/tuning/ is regenerated:
    HSP_TUNABLEOP_OUT=<file> python tools/bench_train_step.py ; python tools/bench_infer.py ; python bench.py"""
    path = path or os.environ.get("HSP_TUNABLEOP_OUT")
    fn = getattr(torch.cuda.tunable, "write_file", None)    # (this torch build appends to the file as it tunes)
    if _enabled and path and fn is not None:
        fn(path)
    return path
