"""average duration per (kernel, grid) of the kernels whose name holds one of the given substrings, from a rocprofv3 kernel trace csv:
python tools/trace_grep.py <kernel_trace.csv> orl_ chunk_fold"""
import collections
import csv
import sys

d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if any(p in r["Kernel_Name"] for p in sys.argv[2:]):
        d[(r["Kernel_Name"][:48], r["Grid_Size_X"], r["Grid_Size_Y"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k_, v in d.items():
    print(f"{k_[0]:48s} grid {k_[1]:>8s}x{k_[2]:<4s} calls {len(v):5d}  avg {sum(v) / len(v) / 1e3:7.1f} us")
