"""idle time between kernels in a rocprofv3 kernel trace: python tools/trace_gaps.py trace.csv [last_k_steps]
Steps are split at concat_rows_kernel dispatches.  Prints, for the last k steps, the step span, the summed kernel time, the summed idle
time (no kernel running), and the largest gaps with the kernels on either side."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "concat_rows_kernel" in r["Kernel_Name"]]
rows = rows[marks[-k - 1]:marks[-1]]
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e3
gaps = []
end = int(rows[0]["End_Timestamp"])
prev = rows[0]
for r in rows[1:]:
    s = int(r["Start_Timestamp"])
    if s > end:
        gaps.append(((s - end) / 1e3, prev["Kernel_Name"][:50], r["Kernel_Name"][:50]))
    if int(r["End_Timestamp"]) > end:
        end = int(r["End_Timestamp"]); prev = r
idle = sum(g[0] for g in gaps)
print(f"{k} steps: span {span / k:9.1f} us/step, kernel time {busy / k:9.1f} us/step, idle {idle / k:8.1f} us/step in {len(gaps) / k:.0f} gaps/step")
hist = [0, 0, 0, 0]
for g in gaps:
    hist[0 if g[0] < 2 else 1 if g[0] < 10 else 2 if g[0] < 50 else 3] += g[0]
print("idle by gap size per step: <2 us %.1f, 2-10 us %.1f, 10-50 us %.1f, >50 us %.1f" % tuple(h / k for h in hist))
for g in sorted(gaps, reverse=True)[:25]:
    print(f"  {g[0]:8.1f} us  after {g[1]:50s} before {g[2]}")
