#!/bin/bash
# rocprofv3 kernel stats of the default bench run; run on the GPU box via gpurun. $1 = output tag
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $R/bench.py --steps 50 --warmup 10 > $OUT/bench.log 2>&1
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
cp $f $OUT/kernel_stats.csv
tail -1 $OUT/bench.log | cut -c1-200
python - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
steps = 60 + 3 + 3   # timed + warmup + capture warm-ups (approximate divisor; see bench.py)
print("total kernel ms", tot / 1e6)
for r in rows[:45]:
    print(f'{float(r["TotalDurationNs"])/1e3:10.0f} us  {int(r["Calls"]):6d}  {float(r["AverageNs"])/1e3:8.1f}  {r["Name"][:110]}')
PY
