#!/bin/bash
# per-shape in-graph kernel tables of the fp32 headline step for both GEMM modes (no side runs); run on the GPU box via gpurun
#   tools/prof_modes.sh <tag> [modes...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-x}; shift
MODES=${@:-"library own"}
for m in $MODES; do
  O=$R/gpurun_out/modes_$TAG/$m
  rm -rf $O; mkdir -p $O
  HSP_GEMM=$m rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python $R/bench.py --no-cpu-baseline --no-u3 --no-side > $O/bench.log 2>&1
  tail -1 $O/bench.log | cut -c1-220
  T=$(find $O/st -name '*kernel_trace.csv' | head -1)
  python $R/tools/trace_by_shape.py $T auto > $O/per_shape.txt
  cp $(find $O/st -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
  rm -rf $O/st
done
