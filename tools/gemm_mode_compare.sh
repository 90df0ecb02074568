#!/bin/bash
# per-shape GEMM kernel times of the step, own kernels vs BLAS library:  tools/gemm_mode_compare.sh   (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/gmc; rm -rf $O; mkdir -p $O
for m in own library; do
  HSP_GEMM=$m rocprofv3 --kernel-trace --output-format csv -d $O/$m -- python $R/bench.py --no-cpu-baseline --no-u3 "$@" > $O/$m.log 2>&1
  tail -1 $O/$m.log | cut -c1-220
  T=$(find $O/$m -name '*kernel_trace.csv' | head -1)
  python $R/tools/trace_by_shape.py $T auto gemm_rows,Cijk,wgrad > $O/$m.txt
  awk '{s+=$NF} END{print "   GEMM-family us/step:", s}' $O/$m.txt
  rm -rf $O/$m
done
