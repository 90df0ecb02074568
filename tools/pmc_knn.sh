#!/bin/bash
# PMC passes over the isolated feature-KNN driver (tools/run_knn_feat.py); run on the GPU box via gpurun.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_knn
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
           "SQ_WAVES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python $R/tools/run_knn_feat.py > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  echo "== pass $i: $f"
  python - "$f" <<'PY'
import csv, sys, collections
if len(sys.argv) < 2 or not sys.argv[1]:
    sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen=set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:40] + " g" + r.get("Grid_Size", "")
    if "knn_feat" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key=(k, r["Dispatch_Id"])
    if key not in seen: seen.add(key); n[k]+=1
for k in acc:
    print(k, "dispatches", n[k])
    for c, v in acc[k].items(): print(f"   {c:32s} {v / n[k]:16.0f}")
PY
done
