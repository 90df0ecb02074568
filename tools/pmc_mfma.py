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


def main(raw_csv, derived_csv=None):
    raw, nr = load(raw_csv)
    der, nd = load(derived_csv)
    rows = []
    for k, c in raw.items():
        d = len(nr[k])
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / d
        mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / d
        if mf <= 0:
            continue
        dd = der.get(k, {})
        rows.append((mf, k, d, gui, c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) / d,
                     dd.get("MfmaUtil", float("nan")) / max(len(nd.get(k, [1])), 1),
                     dd.get("VALUBusy", float("nan")) / max(len(nd.get(k, [1])), 1)))
    rows.sort(reverse=True)
    print(f"{'kernel':58s} {'grid':>9s} {'disp':>4s} {'GUI_ACTIVE':>11s} {'MFMA_BUSY':>12s} {'busy/(GUI*128)':>15s} {'MfmaUtil':>8s} {'VALUBusy':>8s}")
    for mf, (name, grid), d, gui, mops, mu, vb in rows[:40]:
        frac = mf / (gui * 128) if gui else float("nan")
        print(f"{name[:58]:58s} {grid:>9s} {d:4d} {gui:11.0f} {mf:12.0f} {100 * frac:14.1f}% {mu:8.1f} {vb:8.1f}")


if __name__ == "__main__":
    main(*sys.argv[1:3])
