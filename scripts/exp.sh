echo "== torchrun nproc 1"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline 2>gpurun_out/tr.err | cut -c1-700; tail -3 gpurun_out/tr.err
echo "== webvid"
python bench.py --workload webvid --steps 5 --warmup 2 --cpu-seconds 5 2>gpurun_out/wv.err | cut -c1-2500; grep -E "bootstrap|L_pq=" gpurun_out/wv.err | tail -8
echo "== sift"
python bench.py --workload sift --steps 5 --warmup 2 --cpu-seconds 5 2>gpurun_out/sf.err | cut -c1-2500; grep -E "bootstrap|L_pq=" gpurun_out/sf.err | tail -8
