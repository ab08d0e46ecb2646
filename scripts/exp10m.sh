for L in 20 30 40 60; do python bench.py --steps 2 --warmup 1 --no-cpu-baseline --L $L --tune 2>&1 | grep tune | grep -E "entries=(2048|4096|8192|16384)"; done
