"""the kernels of ONE step in launch order: python tools/trace_step_sequence.py trace.csv [marker]   (marker: first kernel of a step)"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "concat_rows_kernel"
st = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
seg = rows[st[-2]:st[-1]]
for i, r in enumerate(seg):
    n = re.sub(r"^void ", "", r["Kernel_Name"])
    n = re.sub(r"\(.*", "", n)[:70]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{i:4d} {d:8.1f} us  {n}  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}")
