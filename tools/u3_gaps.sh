cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d /tmp/u3t -- python $R/tools/u3_only.py 8 4 > /tmp/u3.log 2>&1
tail -1 /tmp/u3.log
T=$(find /tmp/u3t -name '*kernel_trace.csv' | head -1)
python $R/tools/trace_gaps.py $T 4
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline --no-u3 --no-side > /tmp/k.log 2>&1
T=$(find /tmp/kt -name '*kernel_trace.csv' | head -1)
python $R/tools/trace_gaps.py $T 5 | head -8
