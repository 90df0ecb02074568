"""(experiment: tools/experiments/README.md -- needs rfconv_mfma.hip built into libhsp.so)
GPU box: the matrix-core schedule of the receptive-field forward (csrc/rfconv_mfma.hip) against the VALU schedule
(HSP_RF_MFMA=0): out / winning rows / winners' support values bit for bit, then event timings of both."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hs_pose_amd import ops, ops_bf16
from hs_pose_amd._lib import lib
from hs_pose_amd.ops import _p, _run, _stream

dev = torch.device("cuda:0")


def run(surface, dt, xyz, idx, dirs, fm, S, want_fwin):
    B, N, k = idx.shape
    SC = dirs.shape[1]
    C = SC // S
    out = torch.zeros(B, N, C, dtype=dt, device=dev)
    arg = torch.zeros(B, N, SC, dtype=torch.uint16, device=dev)
    fwin = torch.zeros(B, N, SC, dtype=dt, device=dev) if (want_fwin and not surface) else None
    sfx = "_bf16" if dt == torch.bfloat16 else ""
    if surface:
        _run("hsp_rf_surface_fwd" + sfx, (_p(xyz), _p(idx), _p(dirs), B, N, k, S, C, _p(out), _p(arg), _stream()))
    else:
        _run("hsp_rf_conv_fwd" + sfx, (_p(xyz), _p(idx), _p(dirs), _p(fm), B, N, k, S, C, _p(out), _p(arg), _p(fwin), _stream()))
    return out, arg, fwin


def case(surface, dt, B, N, C, k, S, want_fwin=True, time_it=False, dup=False):
    g = torch.Generator().manual_seed(B * 1000 + N + C + k)
    xyz = (torch.randn(B, N, 3, generator=g) * 0.05).to(dev)
    if dup:                                                     # duplicated points: zero-length directions, exact ties
        xyz[:, N // 2:] = xyz[:, :N - N // 2]
    feat = torch.relu(torch.randn(B, N, 16, generator=g)).to(dev)
    idx = ops.knn(xyz if surface else feat, k)
    dirs = torch.randn(3, S * C, generator=g).to(dev)
    fm = None if surface else torch.randn(B, N, (S + 1) * C, generator=g).to(dev).to(dt)
    res = {}
    for mode in ("1", "0"):
        os.environ["HSP_RF_MFMA"] = mode
        res[mode] = run(surface, dt, xyz, idx, dirs, fm, S, want_fwin)
    torch.cuda.synchronize()
    ok = True
    for a, b, nm in zip(res["1"], res["0"], ("out", "argrow", "fwin")):
        if a is None:
            continue
        if nm == "argrow":
            a, b = a.view(torch.int16), b.view(torch.int16)
        if not torch.equal(a, b):
            ok = False
            d = (a.float() - b.float()).abs()
            print(f"   MISMATCH {nm}: {int((d > 0).sum())} of {d.numel()} elements, max {float(d.max()):.3e}")
    tag = f"{'surface' if surface else 'conv'} {str(dt)[6:]} B{B} N{N} C{C} k{k} S{S} fwin{int(want_fwin)}{' dup' if dup else ''}"
    line = f"{'ok  ' if ok else 'FAIL'} {tag}"
    if time_it:
        for mode in ("1", "0"):
            os.environ["HSP_RF_MFMA"] = mode
            for _ in range(5):
                run(surface, dt, xyz, idx, dirs, fm, S, want_fwin)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                run(surface, dt, xyz, idx, dirs, fm, S, want_fwin)
            e1.record(); torch.cuda.synchronize()
            line += f"   {'mfma' if mode == '1' else 'valu'} {1000 * e0.elapsed_time(e1) / 30:.1f} us"
    print(line, flush=True)
    os.environ["HSP_RF_MFMA"] = "1"
    return ok


if __name__ == "__main__":
    f32, bf = torch.float32, torch.bfloat16
    allok = True
    T = "--time" in sys.argv
    for args in [
        (False, f32, 16, 1028, 128, 20, 7), (False, f32, 16, 257, 256, 20, 7), (False, f32, 16, 64, 512, 8, 7),
        (True, f32, 16, 1028, 128, 20, 7),
        (False, f32, 2, 256, 128, 20, 7), (False, f32, 1, 1028, 128, 20, 7), (False, f32, 5, 333, 128, 20, 7),
        (False, f32, 3, 100, 32, 8, 3), (False, f32, 3, 100, 64, 5, 3), (False, f32, 2, 64, 32, 2, 7),
        (False, f32, 2, 90, 32, 1, 2), (False, f32, 2, 200, 32, 32, 2), (False, f32, 2, 200, 64, 27, 3),
        (False, f32, 2, 200, 64, 13, 3), (True, f32, 2, 256, 128, 20, 7), (True, f32, 3, 100, 32, 5, 3),
        (True, f32, 1, 1028, 128, 20, 7), (True, f32, 2, 16, 128, 2, 7),
        (False, bf, 8, 4096, 128, 20, 7), (False, bf, 8, 1024, 256, 20, 7), (False, bf, 8, 256, 512, 20, 7),
        (True, bf, 8, 4096, 128, 20, 7), (False, bf, 2, 100, 64, 5, 3),
    ]:
        allok &= case(*args, time_it=T)
    allok &= case(False, f32, 4, 300, 128, 20, 7, want_fwin=False)
    allok &= case(False, f32, 4, 300, 128, 20, 7, dup=True)
    allok &= case(True, f32, 4, 300, 128, 20, 7, dup=True)
    print("ALL OK" if allok else "SOME FAILED")
    sys.exit(0 if allok else 1)
