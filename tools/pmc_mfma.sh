#!/bin/bash
# MFMA / VALU utilisation of the bench's kernels: rocprofv3 PMC passes (--kernel-trace only) over an eager run of bench.py.
# Run on the GPU box via gpurun; results under gpurun_out/pmc_mfma (summary: mfma_util.txt).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_mfma
rm -rf $O; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -i -B1 -A4 "MfmaUtil\|VALUBusy" | head -40 > $O/derived_defs.txt
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" "MfmaUtil VALUBusy" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python $R/bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline --no-u3 ${PMC_BENCH_ARGS} > $O/p$i.log 2>&1
  f=$(find $O/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp "$f" $O/pmc_pass$i.csv
  rm -rf $O/p$i
done
python $R/tools/pmc_mfma.py $O/pmc_pass1.csv $O/pmc_pass2.csv $O/pmc_pass3.csv > $O/mfma_util.txt 2>&1
head -40 $O/mfma_util.txt
