"""The reference's evaluation run on FILES at the benchmark size, through this repo's C++ harness (own counterpart of
tests/test_search_roargraph.cpp): T2I-10M shape written as base.fbin (8 GB) / query.fbin / groundtruth.ibin / roar.index (bench.py's default graph),
then `test_search_roargraph --L_pq 55 ...` -- the drop-in path end to end (file load, host buffers, one launch per L_pq).
Prints the harness's own stdout (QPS column = wall clock of the call, results on the host)."""
import argparse, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mysteryann_amd import fileio, index as mi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--dir", default="/tmp/roar_files")
ap.add_argument("--L", type=int, nargs="+", default=[55, 55, 100, 500])
ap.add_argument("--gpus", type=str, default="")
args = ap.parse_args()
os.makedirs(args.dir, exist_ok=True)
n, d, nq, k = args.n, 200, 10000, 10
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _graph import t2i_index
idx, base, queries = t2i_index(n, nq, "reference", 0, "/tmp/r2_graph.index")
gi, gd = idx.ExactKnnDevice(queries, 100)
P = {x: os.path.join(args.dir, x) for x in ("base.fbin", "query.fbin", "gt.ibin", "roar.index", "eval.csv")}
t0 = time.time()
with open(P["base.fbin"], "wb") as f:
    np.array([n, d], np.uint32).tofile(f)
    for i in range(0, n, 1 << 20):
        base[i:i + (1 << 20)].cpu().numpy().tofile(f)
fileio.write_fbin(P["query.fbin"], queries.cpu().numpy())
fileio.write_gt(P["gt.ibin"], gi.cpu().numpy().view(np.uint32), -gd.cpu().numpy())
idx.SaveProjectionGraph(P["roar.index"])
print(f"# files written in {time.time() - t0:.0f}s: base {os.path.getsize(P['base.fbin'])} B, index {os.path.getsize(P['roar.index'])} B", flush=True)
idx.close(); del base; torch.cuda.empty_cache()
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mysteryann_amd", "csrc", "test_search_roargraph")
cmd = [exe, "--data_type", "float", "--dist", "ip", "--base_data_path", P["base.fbin"], "--query_path", P["query.fbin"], "--gt_path",
       P["gt.ibin"], "--projection_index_save_path", P["roar.index"], "--k", "10", "--evaluation_save_path", P["eval.csv"], "--L_pq"] + [str(x) for x in args.L]
if args.gpus:
    cmd += ["--devices", args.gpus]
t0 = time.time()
out = subprocess.run(cmd, capture_output=True, text=True)
print(out.stdout, out.stderr[-2000:], f"# harness wall {time.time() - t0:.0f}s (file load + upload + {len(args.L)} timed calls)", sep="\n")
