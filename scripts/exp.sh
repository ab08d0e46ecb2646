for L in 150 300 500 1000 1500; do for v in lds exact; do
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --L $L --visited $v 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('L',r['config']['L_pq'],r['config']['visited'],'recall',round(r['config']['recall_at_10'],4),'kernel_ms',round(r['roofline']['kernel_ms'],2),'frac',round(r['roofline']['frac'],3),'grid',r['roofline']['grid'],'lds',r['roofline']['lds_bytes'],'cmps',int(r['roofline']['lds_mode_cmps']))"
done; done
