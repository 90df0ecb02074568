#!/bin/bash
# rocprofv3 kernel stats of the eager inference loop (tools/bench_infer.py); run on the GPU box via gpurun.  $1 = tag, $2 = instances
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_infer_$1
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $R/tools/bench_infer.py --images 40 --instances ${2:-4} > $OUT/bench.log 2>&1
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
cp $f $OUT/kernel_stats.csv
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
python - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6, " (20 warm-up + 80 timed images, eager + graph)")
for r in rows[:32]:
    print(f'{float(r["TotalDurationNs"])/1e3:10.0f} us  {int(r["Calls"]):6d}  {float(r["AverageNs"])/1e3:8.1f}  {r["Name"][:100]}')
PY
