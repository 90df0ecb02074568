#!/bin/bash
# per-shape kernel table of the default bench (graph replay) only; run on the GPU box via gpurun. $1 = output tag
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/shape_$1
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --no-cpu-baseline --no-u3 --no-side > $O/bench.log 2>&1
tail -1 $O/bench.log | cut -c1-200
T=$(find $O/stats -name '*kernel_trace.csv' | head -1)
python $R/tools/trace_by_shape.py $T auto > $O/k_per_shape_kernel_us.txt
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/k_graph_kernel_stats.csv
rm -rf $O/stats
