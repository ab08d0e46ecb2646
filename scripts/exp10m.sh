run() { echo "== $*"; python bench.py --sweep --max-L 150 --no-cpu-baseline --steps 2 --warmup 1 "$@" 2>&1 | grep -E "stage 2|bootstrap graph|L_pq= *(30|40|50|60|65|70|75|80|100|150) " | sed 's/\[rank 0\] //'; }
run --two-stage 1
run --two-stage 1 --stage2-L 256 --stage2-Msq 63 --stage2-Mpj 35 --stage2-supply 24 --stage2-maxdeg 70
