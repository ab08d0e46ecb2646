#!/bin/bash
# Round 5: experiments on the kNN pass kernel (one 65 536-query block of the builder's lists each): forms side by side (ROAR_KNN_FORM),
# per-phase cycle counts (ROAR_KNN_PROF), the last pass by device events (ROAR_KNN_TIME), timing with nothing admitted
# (ROAR_KNN_DEBUG=1: results are wrong, the time is the pipeline's).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5
{
for v in ${VARIANTS:-"ROAR_KNN_FORM=0,ROAR_KNN_TIME=1" "ROAR_KNN_FORM=3,ROAR_KNN_TIME=1" "ROAR_KNN_FORM=3,ROAR_KNN_DEBUG=1,ROAR_KNN_TIME=1"}; do
  echo "== $v"
  env ${v//,/ } timeout 600 python3 scripts/knn_profile.py 2>&1 | grep -v amdgpu.ids | tail -4
done
if [ -n "${TESTS:-}" ]; then ROAR_KNN_FORM=$TESTS timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "knn" 2>&1 | tail -5; fi
} 2>&1 | tee gpurun_out/r5/knn_forms.txt
