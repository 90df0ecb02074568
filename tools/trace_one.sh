#!/bin/bash
# kernel trace of one python script on the GPU box: per (kernel, grid, workgroup) count / average / minimum duration in us.
#   tools/trace_one.sh <tag> <script.py> [args...]      (via gpurun; output under gpurun_out/trace_<tag>/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
O=$R/gpurun_out/trace_$TAG
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python "$@" > $O/run.log 2>&1
f=$(find $O -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY' | tee $O/summary.txt
import csv, sys, collections
d = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    key = (r["Kernel_Name"][:70], r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", r.get("Workgroup_Size")), r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""))
    d.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    print(f"{k[0]:70s} grid {k[1]:>8s} wg {k[2]:>4s} lds {k[3]:>7s} vgpr {k[4]:>4s}  n {len(v):4d}  avg {sum(v)/len(v):8.1f}  min {min(v):8.1f}")
PY
rm -f $f
