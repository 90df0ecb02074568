"""where the time of a replayed step goes, from a rocprofv3 kernel trace CSV: python tools/trace_gaps.py trace.csv [marker substring]
Splits the trace into steps at each dispatch of the marker kernel (default: the first kernel of HSPose's forward that
appears once per step), then prints for the last steps: span, summed kernel time, idle time, and the largest gaps with the
kernels on either side."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "loss_prep_kernel"
starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
print("dispatches", len(rows), "steps", len(starts))
for a, b in list(zip(starts[:-1], starts[1:]))[-3:]:
    seg = rows[a:b]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    gaps = []
    for x, y in zip(seg[:-1], seg[1:]):
        g = int(y["Start_Timestamp"]) - int(x["End_Timestamp"])
        gaps.append((g, x["Kernel_Name"][:50], y["Kernel_Name"][:50]))
    pos = sum(g for g, _, _ in gaps if g > 0)
    print(f"step: {len(seg)} kernels, span {(t1 - t0) / 1e6:.2f} ms, kernel time {busy / 1e6:.2f} ms, idle between kernels {pos / 1e6:.2f} ms,"
          f" median gap {sorted(g for g, _, _ in gaps)[len(gaps) // 2] / 1e3:.1f} us")
    for g, x, y in sorted(gaps, reverse=True)[:6]:
        print(f"    gap {g / 1e3:8.1f} us  after {x}  before {y}")
