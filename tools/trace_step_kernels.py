"""kernels of ONE step of a traced run, in launch order with counts: python tools/trace_step_kernels.py trace.csv [marker]
(a step = the dispatches between the last two dispatches of the marker kernel)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "loss_prep_kernel"
st = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
seg = rows[st[-2]:st[-1]]
cnt = collections.Counter(); dur = collections.Counter()
for r in seg:
    n = r["Kernel_Name"][:110]
    cnt[n] += 1; dur[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(len(seg), "kernels,", round(sum(dur.values()) / 1e3, 2), "ms of kernel time")
for n, c in sorted(cnt.items(), key=lambda kv: -dur[kv[0]]):
    if len(sys.argv) > 3 and sys.argv[3] == "aten" and ("hsp::" in n or n.startswith("Cijk") or "gather_rows" in n):
        continue
    print(f"{c:4d} x {dur[n]:8.1f} us  {n}")
