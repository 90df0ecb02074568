#!/bin/bash
# generic PMC passes: tools/pmc_run.sh <kernel-name substring> <out tag> -- <command ...>     (run on the GPU box via gpurun)
# prints per-kernel (name, grid) averages of every counter; raw CSVs under gpurun_out/pmc_<tag>/
FILTER=$1; TAG=$2; shift 3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
           "SQ_WAVES SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- "$@" > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  python - "$f" "$FILTER" <<'PY'
import csv, sys, collections
if len(sys.argv) < 3 or not sys.argv[1]:
    sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen=set()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] not in r["Kernel_Name"]: continue
    k = r["Kernel_Name"].split("(")[0][-70:] + " g" + r.get("Grid_Size", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key=(k, r["Dispatch_Id"])
    if key not in seen: seen.add(key); n[k]+=1
for k in acc:
    print(k, "dispatches", n[k])
    for c, v in acc[k].items(): print(f"   {c:32s} {v / n[k]:16.0f}")
PY
done
