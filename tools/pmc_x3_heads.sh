#!/bin/bash
# memory-path counters of gemm_x3 on the deep-K head shapes (tools/time_x3_tall.py); run on the GPU box via gpurun -> gpurun_out/pmc_x3
# NOTE (round 5): only the two TCC passes finish -- a pass with TA_* / TCP_* counters did not complete within 15 minutes on this
# pool (twice), the SQ set is in tools/pmc_x3_one.sh.  The loop below stops after the TCC passes for that reason.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_x3
rm -rf $O; mkdir -p $O
i=0
for set in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUSY_avr TCC_TAG_STALL_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python $R/tools/x3_one.py > $O/p$i.log 2>&1
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
        key = (r["Kernel_Name"][:60], r["Grid_Size"])
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, d in acc.items():
    print(key)
    for c, v in sorted(d.items()):
        print(f"   {c:44s} avg {sum(v) / len(v):16.1f}  (n={len(v)})")
PY
