run() { echo "== $*"; python bench.py --n 50000 --nq 1000 --learn 50000 --sweep --max-L 200 --no-cpu-baseline --steps 2 --warmup 1 "$@" 2>&1 | grep -E "bootstrap graph|L_pq= *(10|20|50|100|200) " | sed 's/\[rank 0\] //'; }
run
run --exact-knn 50000 --build-rounds 0
run --exact-knn 50000 --build-rounds 0 --build-Msq 63 --build-Mpj 32 --build-supply 24 --build-C3 60 --build-supplyL 256
run --exact-knn 50000 --build-rounds 0 --build-Msq 63 --build-Mpj 35 --build-M 8 --build-basecap 24 --build-supply 35 --build-C3 63 --build-supplyL 500 --build-maxdeg 105
run --exact-knn 50000 --build-rounds 1 --build-Msq 63 --build-Mpj 32
