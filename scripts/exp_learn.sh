run() { echo "== $*"; python bench.py --sweep --max-L 40 --no-cpu-baseline --steps 2 --warmup 1 "$@" 2>&1 | grep -E "bootstrap graph|L_pq= *(25|30|35) " | sed 's/\[rank 0\] //'; }
run --build-maxdeg 200
run --build-maxdeg 208
run --build-maxdeg 176
