#!/bin/bash
# regenerate the judged artifacts under gpurun_out/refresh (then copied into profiles/<round>/ by hand):
#   kernel stats of the default bench run (rocprofv3 --kernel-trace --stats), the bench line, per-shape table,
#   PMC traffic (separate FETCH_SIZE / WRITE_SIZE passes), eager per-kernel event breakdown.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
python $R/bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --no-cpu-baseline > $O/stats_bench.log 2>&1
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/graph_kernel_stats.csv
T=$(find $O/stats -name '*kernel_trace.csv' | head -1)
python $R/tools/trace_by_shape.py $T auto > $O/per_shape_kernel_us.txt
python $R/bench.py --steps 20 --warmup 5 --breakdown --no-cpu-baseline > $O/breakdown_eager_events.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline > $O/pmc_$c.log 2>&1
  cp $(find $O/pmc_$c -name '*counter_collection.csv' | head -1) $O/pmc_$c.csv
done
python $R/tools/pmc_traffic.py $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv $O/traffic.json > /dev/null
rm -rf $O/stats $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
ls -la $O
