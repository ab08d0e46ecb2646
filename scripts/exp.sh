for cfg in "--build-rounds 0 --learn 1000000 --exact-knn" "--build-rounds 2 --learn 1000000 --exact-knn" "--build-rounds 0 --learn 1000000 --exact-knn --build-Msq 64 --build-Mpj 32 --build-maxdeg 96" "--build-rounds 5 --learn 1000000 --build-Llearn 128 --build-Msq 48 --build-Mpj 24 --build-maxdeg 96"; do
echo "== $cfg"
python bench.py --n 1000000 --steps 3 --warmup 1 --no-cpu-baseline --sweep $cfg 2>&1 | grep -E "exact learn|bootstrap graph|L_pq= *(10|20|30|50|100|200|500) " | cut -c1-400
done
