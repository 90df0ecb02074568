#!/bin/bash
# HBM traffic of the bench's kernels: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) over an
# eager run of bench.py.  Run on the GPU box via gpurun; $1 = output tag.  Then tools/pmc_traffic.py.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  OUT=$R/gpurun_out/pmc_$1_$c
  mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT -- python $R/bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline > $OUT/bench.log 2>&1
  f=$(find $OUT -name '*counter_collection.csv' | head -1)
  cp "$f" $R/gpurun_out/pmc_$1_$c.csv
  tail -1 $OUT/bench.log | cut -c1-120
done
python $R/tools/pmc_traffic.py $R/gpurun_out/pmc_$1_FETCH_SIZE.csv $R/gpurun_out/pmc_$1_WRITE_SIZE.csv $R/gpurun_out/traffic_$1.json
