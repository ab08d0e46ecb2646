#!/bin/bash
# Round 6: how much of a hop's OWN time does a launch's rate hang on?  Variant libraries (built beforehand, here or on the box:
#   for n in 16 32 64; do scripts/build_variant.sh libroargraph_amd_pad$n.so -DROAR_HOP_PAD=$n; done)
# make every hop's wave sit out n x 64 more cycles where the merge is, issuing nothing.  Per library: 10 000-query launches at L_pq 50
# (4 096 tags) and 1 000 (2 560 tags), one at a time, and the stream at L_pq 50.  If the kernel time rises by (pad / hop time) the
# launch is bound by its waves' own latency chains (a faster merge would pay in proportion); if it does not move, by the memory system.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6
{
for lib in "" _pad16 _pad32 _pad64; do
  f=$PWD/mysteryann_amd/csrc/libroargraph_amd$lib.so
  [ -f $f ] || continue
  echo "== library libroargraph_amd$lib.so"
  ROAR_AMD_LIB=$f timeout 600 python3 scripts/longpool_probe.py --L 50 --tables 4096 --no-exact 2>&1 | grep -v amdgpu.ids | grep "L_pq\|graph"
  ROAR_AMD_LIB=$f timeout 600 python3 scripts/longpool_probe.py --L 1000 --tables 2560 --no-exact 2>&1 | grep -v amdgpu.ids | grep "L_pq"
  ROAR_AMD_LIB=$f ROUNDS=1 timeout 600 python3 scripts/table_regime_probe.py 4096 2>&1 | grep "round"
done
} 2>&1 | tee gpurun_out/r6/hop_pad_probe.txt
