#!/bin/bash
# rocprofv3 kernel statistics of the eager full training step (tools/bench_train_step.py); run on the GPU box via gpurun
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_u3
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tools/bench_train_step.py --steps 10 --warmup 3 ${U3_MODE:---no-graph} > $O/bench.log 2>&1
tail -1 $O/bench.log
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/u3_kernel_stats.csv
python - $O/u3_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 13
tot = sum(int(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print(f"kernels/step {calls / steps:.0f}   kernel time/step {tot / steps / 1e6:.2f} ms   mean {tot / calls / 1e3:.2f} us")
fam = {"hsp": 0, "gemm": 0, "aten": 0}; cnt = dict(fam)
for r in rows:
    n = r["Name"]; k = "hsp" if ("hsp::" in n or n.startswith("gather_rows")) else "gemm" if n.startswith("Cijk") else "aten"
    fam[k] += int(r["TotalDurationNs"]); cnt[k] += int(r["Calls"])
for k in fam: print(f"  {k:5s} {fam[k] / steps / 1e6:7.2f} ms/step  {cnt[k] / steps:7.0f} kernels/step")
for r in rows[:14]: print(f"  {int(r['TotalDurationNs']) / steps / 1e3:8.1f} us/step  x{int(r['Calls']) / steps:6.1f}  {r['Name'][:100]}")
PY
rm -rf $O/stats
