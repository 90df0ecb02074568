#!/bin/bash
# rocprofv3 kernel statistics of the full training step (unit U3: tools/u3_only.py = bench.py::u3_full_step); run on the GPU
# box via gpurun.  Prints the kernels of one step sorted by time (the warm-up's captures are eager passes of the same kernels).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_u3
STEPS=${U3_STEPS:-10}; WARM=${U3_WARMUP:-4}
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tools/u3_only.py $STEPS $WARM > $O/bench.log 2>&1
tail -1 $O/bench.log
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/u3_kernel_stats.csv
T=$(find $O/stats -name '*kernel_trace.csv' | head -1)
python $R/tools/trace_by_shape.py $T last4 > $O/u3_per_shape_kernel_us.txt 2>&1
head -70 $O/u3_per_shape_kernel_us.txt
rm -rf $O/stats
