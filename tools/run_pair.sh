#!/bin/bash
# two concurrent processes of a debug script: tools/dbg_pair.sh script.py
cd $GRAFT_REPO_ROOT
python $1 A > gpurun_out/ra.log 2>&1 &
python $1 B > gpurun_out/rb.log 2>&1
wait
grep "^\[" gpurun_out/ra.log | head -12; grep "^\[" gpurun_out/rb.log | head -12
