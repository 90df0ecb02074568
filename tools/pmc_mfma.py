"""Per-kernel MFMA / VALU utilisation from the rocprofv3 PMC passes of tools/pmc_mfma.sh.

pass 1 (raw): SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_ACTIVE_INST_VALU, SQ_INSTS_VALU_MFMA_MOPS_F32,
GRBM_GUI_ACTIVE;  pass 2 (rocprofv3's derived metrics, gfx94x formulas on gfx950): MfmaUtil, VALUBusy.
busy% = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs): rocprofv3 sums GRBM_GUI_ACTIVE over the 8
XCDs; this reproduces the derived MfmaUtil to within a few per cent (two separate runs).
"""
import collections
import csv
import os
import sys


def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    if not path or not os.path.exists(path):
        return acc, n
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if name.startswith("Cijk_"):
            name = name[:48]
        k = (name, r.get("Grid_Size", ""))
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    return acc, n


def main(raw_csv, derived_csv=None, wait_csv=None):
    """every kernel of the run, longest first (GRBM_GUI_ACTIVE per dispatch = its duration in GPU clocks summed over the 8 XCDs):
    matrix-pipe busy share, VALU busy share (rocprofv3's derived VALUBusy), and the share of the waves' cycles spent waiting on
    an instruction's operands / counters (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, third pass) -- kernels without an MFMA included"""
    raw, nr = load(raw_csv)
    der, nd = load(derived_csv)
    wt, nw = load(wait_csv)
    rows = []
    for k, c in raw.items():
        d = len(nr[k])
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / d
        mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / d
        dd = der.get(k, {})
        ww = wt.get(k, {})
        wait = ww.get("SQ_WAIT_INST_ANY", float("nan")) / ww["SQ_WAVE_CYCLES"] if ww.get("SQ_WAVE_CYCLES") else float("nan")
        valu_per_disp = ww.get("SQ_INSTS_VALU", float("nan")) / max(len(nw.get(k, [1])), 1)
        rows.append((gui * d, k, d, gui, mf,
                     dd.get("MfmaUtil", float("nan")) / max(len(nd.get(k, [1])), 1),
                     dd.get("VALUBusy", float("nan")) / max(len(nd.get(k, [1])), 1), wait, valu_per_disp))
    rows.sort(reverse=True)
    print(f"{'kernel':58s} {'grid':>9s} {'disp':>4s} {'GUI_ACTIVE':>11s} {'MFMA busy/(GUI*128)':>20s} {'MfmaUtil':>8s} {'VALUBusy':>8s} {'wait share':>10s} {'VALU inst/disp':>14s}")
    for _, (name, grid), d, gui, mf, mu, vb, wait, nv in rows[:45]:
        frac = mf / (gui * 128) if gui else float("nan")
        print(f"{name[:58]:58s} {grid:>9s} {d:4d} {gui:11.0f} {100 * frac:19.1f}% {mu:8.1f} {vb:8.1f} {100 * wait:9.1f}% {nv:14.0f}")


if __name__ == "__main__":
    main(*sys.argv[1:4])
