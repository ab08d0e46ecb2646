run() { echo "== $*"; python bench.py --learn 10000000 --sweep --max-L 80 --no-cpu-baseline --steps 2 --warmup 1 "$@" 2>&1 | grep -E "bootstrap graph|L_pq= *(45|50|55|60|65) " | sed 's/\[rank 0\] //'; }
run --build-rounds 3
run --build-Llast 384
run --build-Mpj 28 --build-Msq 56
