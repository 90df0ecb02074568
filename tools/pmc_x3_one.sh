#!/bin/bash
# stall / unit-busy counters of ONE gemm_x3 head product (tools/x3_one.py); run on the GPU box via gpurun -> gpurun_out/pmc_x3_one
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_x3_one
rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python $R/tools/x3_one.py $X3_ONE_ARGS > $O/p$i.log 2>&1
  f=$(find $O/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp "$f" $O/pass$i.csv
  rm -rf $O/p$i
done
python - $O <<'PY'
import csv, sys, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1] + "/pass*.csv")):
    for r in csv.DictReader(open(f)):
        if "gemm_x3_kernel" not in r["Kernel_Name"]:
            continue
        acc[r["Kernel_Name"][:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, d in acc.items():
    print(key)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} {sum(v) / len(v):18.1f}  (n={len(v)})")
PY
