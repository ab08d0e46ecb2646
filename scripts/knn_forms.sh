#!/bin/bash
# Round 5: experiments on the kNN pass kernel (one 65 536-query block of the builder's lists each; scripts/knn_profile.py [N [D]]):
# variants by environment (ROAR_KNN_SPLIT launches per pass, ROAR_KNN_RATIO sample ratio, ROAR_KNN_PROF per-phase cycle counts of the
# instrumented kernel, ROAR_KNN_TIME the last pass by device events, ROAR_KNN_DEBUG=1 nothing admitted -- results are wrong, the time is
# the pipeline's; ROAR_AMD_LIB another build of the library), SHAPES="N,D ..." (default the T2I-10M shape).
# The switches exist only in a library built with -DROAR_KNN_EXPERIMENTS (round 6: the shipped library reads none of them): built here
# as a variant and selected with ROAR_AMD_LIB unless the caller has set one.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5
if [ -z "${ROAR_AMD_LIB:-}" ]; then
  scripts/build_variant.sh libroargraph_amd_knnexp.so -DROAR_KNN_EXPERIMENTS >/dev/null
  export ROAR_AMD_LIB=$PWD/mysteryann_amd/csrc/libroargraph_amd_knnexp.so
fi
{
for shape in ${SHAPES:-10000000,200}; do
for v in ${VARIANTS:-"ROAR_KNN_SPLIT=0"}; do
  echo "== ${shape} $v"
  env ${v//,/ } timeout 600 python3 scripts/knn_profile.py ${shape//,/ } 2>&1 | grep -v amdgpu.ids | tail -4
done
done
if [ -n "${TESTS:-}" ]; then timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "knn" 2>&1 | tail -5; fi
} 2>&1 | tee gpurun_out/r5/knn_forms.txt
