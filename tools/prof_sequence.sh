#!/bin/bash
# kernel sequence of one graph-replayed step (launch order, durations) + the per-shape table; run on the GPU box via gpurun. $1 = tag
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/seq_$1
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python $R/bench.py --no-cpu-baseline --no-u3 --no-side --steps 20 --warmup 5 > $OUT/bench.log 2>&1
T=$(find $OUT/t -name '*kernel_trace.csv' | head -1)
python $R/tools/trace_step_sequence.py $T > $OUT/sequence.txt
python $R/tools/trace_by_shape.py $T last3 > $OUT/per_shape.txt
tail -1 $OUT/bench.log | cut -c1-160
wc -l $OUT/sequence.txt
awk '{ if ($2 < 6.5) { n++; s += $2 } } END { print "launches < 6.5 us:", n, "sum", s, "us" }' $OUT/sequence.txt
rm -rf $OUT/t
