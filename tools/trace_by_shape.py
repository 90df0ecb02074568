"""per-(kernel, grid) duration table from a rocprofv3 kernel trace CSV:
    python tools/trace_by_shape.py trace.csv <steps | auto> [filter]
'auto': the number of steps is the number of concat_rows_kernel dispatches (one per forward)."""
import csv, collections, sys
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
if sys.argv[2].startswith("last"):
    # "last<k>": only the final k steps (after warm-up / library tuning), split at the concat_rows_kernel dispatches
    k = int(sys.argv[2][4:] or 3)
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "concat_rows_kernel" in r["Kernel_Name"]]
    rows = rows[marks[-k - 1]:marks[-1]]
    sys.argv[2] = str(k)
steps = float(sum("concat_rows_kernel" in r["Kernel_Name"] for r in rows)) if sys.argv[2] == "auto" else float(sys.argv[2])
flt = sys.argv[3].split(",") if len(sys.argv) > 3 else None
acc = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if flt and not any(x in n for x in flt):
        continue
    key = (n[:60], f'{r["Grid_Size_X"]}x{r["Grid_Size_Y"]}', r["Workgroup_Size_X"])
    acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[0]:60s} grid {k[1]:>12s} wg {k[2]:>5s} calls/step {len(v)/steps:4.1f} avg {sum(v)/len(v):7.1f} us  per-step {sum(v)/steps:7.1f}")
