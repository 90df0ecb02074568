cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in base fl8 fl16; do
  if [ $v = base ]; then unset HSP_LIB; else export HSP_LIB=$R/build_tmp/libhsp_$v.so; fi
  echo "== $v"
  rocprofv3 --kernel-trace --output-format csv -d /tmp/o_$v -- python $R/tools/time_scatter.py > /dev/null 2>&1
  T=$(find /tmp/o_$v -name '*kernel_trace.csv' | head -1)
  python $R/tools/trace_grep.py $T scatter_tile
  python $R/bench.py --no-cpu-baseline --no-u3 --no-side | cut -c1-110
done
