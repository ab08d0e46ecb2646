for cfg in "--build-M 16 --build-basecap 64 --build-maxdeg 128 --learn 2500000 --build-rounds 5" "--build-M 20 --build-basecap 80 --build-maxdeg 128 --learn 2500000 --build-rounds 3" "--build-M 16 --build-basecap 64 --build-maxdeg 128 --learn 5000000 --build-rounds 3 --build-Msq 64 --build-Mpj 32" ; do
echo "== $cfg"
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --sweep --max-L 200 $cfg 2>&1 | grep -E "exact learn|bootstrap graph|L_pq= *(10|20|30|40|50|60|70|80|100|150|200) " | cut -c1-400
done
