#!/bin/bash
# ablation builds of the deep gemm_x3 pipeline (X3_ABL bits: 1 no MFMA, 2 no global loads, 4 no split / LDS writes); results are WRONG by
# construction, only the times mean something.  Build here (no GPU needed): bash tools/experiments/x3_deep_ablate.sh (after applying an X3_ABL build of gemm_x3_deep_pipeline_wm4.patch) build ; run on the GPU box: bash tools/experiments/x3_deep_ablate.sh (after applying gemm_x3_deep_pipeline.patch) run
R=$(cd $(dirname $0)/../.. && pwd)
C=$R/hs_pose_amd/csrc
mkdir -p $R/build_tmp
if [ "$1" = build ]; then
  for a in 1 2 4 6 7; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -I$R/include -DX3_ABL=$a -c $C/gemm_x3.hip -o $R/build_tmp/gemm_x3_abl$a.o &
  done
  wait
  for a in 1 2 4 6 7; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $R/build_tmp/gemm_x3_abl$a.o $(ls $C/*.o | grep -v "/gemm_x3.o") -o $R/build_tmp/libhsp_x3abl$a.so
  done
else
  for a in 1 2 4 6 7; do
    echo "X3_ABL=$a"; HSP_LIB=$R/build_tmp/libhsp_x3abl$a.so python $R/tools/time_x3_tall.py 2>&1 | grep M16448 | head -1
  done
fi
