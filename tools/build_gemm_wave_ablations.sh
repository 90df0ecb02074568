#!/bin/bash
# profiling builds of libhsp.so with parts of gemm_wave.hip removed (GW_ABLATE = 1 no MFMA, 2 no operand loads, 3 no stores)
# -> build_tmp/libhsp_ab{1,2,3}.so; select with HSP_LIB=build_tmp/libhsp_abN.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/hs_pose_amd/csrc
mkdir -p $R/build_tmp
make -C $C -s -j8
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -I$R/include -I$C \
      -Wall -Wno-unused-function -DGW_ABLATE=$m -c $C/gemm_wave.hip -o $R/build_tmp/gemm_wave_ab$m.o
  OBJS=$(ls $C/*.o | grep -v gemm_wave.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $R/build_tmp/gemm_wave_ab$m.o -o $R/build_tmp/libhsp_ab$m.so
done
