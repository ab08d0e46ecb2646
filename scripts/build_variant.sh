#!/bin/bash
# A second build of the product library with extra compiler flags, for A/B measurements in one box session:
#   scripts/build_variant.sh libroargraph_amd_merge1.so -DROAR_MERGE_2WAY=0
#   ROAR_AMD_LIB=$PWD/mysteryann_amd/csrc/libroargraph_amd_merge1.so python scripts/lat_probe.py ...
set -e
out=$1; shift
here=$(cd "$(dirname "$0")/../mysteryann_amd/csrc" && pwd)
tmp=$(mktemp -d /tmp/roar_variant.XXXX)
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $*"
pids=""
for f in roar_kernels.hip roar_search_lds_ip.hip roar_search_lds_l2.hip roar_search_exact_ip.hip roar_search_exact_l2.hip roar_build.hip roar_rgbuild.hip roar_knn.hip roar_knn_bf16.hip roar_hub.hip; do
  /opt/rocm/bin/hipcc $flags -c $here/$f -o $tmp/${f%.hip}.o & pids="$pids $!"
done
for f in $(cd $here && ls roar_*.cpp); do
  /opt/rocm/bin/hipcc $flags -x hip -c $here/$f -o $tmp/${f%.cpp}.o & pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $here/$out $tmp/*.o -L/opt/rocm/lib -lrccl
rm -rf $tmp
echo "built $here/$out with: $*"
