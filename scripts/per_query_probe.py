"""The reference harness's loop shape on the GPU: `test_search_roargraph --per_query -T n` (one SearchRoarGraph
call per query from n OpenMP threads, calls combined inside the library) against the batch call, 1 M x 200 IP."""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mysteryann_amd import fileio, index as mi, synth
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "mysteryann_amd", "csrc")
n, nq, d = 1_000_000, 10_000, 200
tmp = tempfile.mkdtemp(prefix="pq_")
base, q, learn = synth.t2i_torch(n, nq, d, n_learn=n)
idx = mi.IndexBipartite(d, n, mi.INNER_PRODUCT, None); idx.SetBaseData(base)
idx.BuildBootstrapGraph(learn=learn, max_degree=200); torch.cuda.synchronize()
P = {k: os.path.join(tmp, k) for k in ("base.fbin", "query.fbin", "gt.ibin", "roar.index", "eval.csv")}
fileio.write_fbin(P["base.fbin"], base[:, :d].cpu().numpy()); fileio.write_fbin(P["query.fbin"], q[:, :d].cpu().numpy())
idx.SaveProjectionGraph(P["roar.index"]); idx.close(); del idx, base, learn; torch.cuda.empty_cache()
subprocess.check_call([os.path.join(csrc, "compute_knn"), "--dist", "ip", "--base_data_path", P["base.fbin"], "--query_path",
                       P["query.fbin"], "--out_path", P["gt.ibin"], "--K", "100", "--with_dists"], stdout=subprocess.DEVNULL)
common = [os.path.join(csrc, "test_search_roargraph"), "--data_type", "float", "--dist", "ip", "--base_data_path", P["base.fbin"],
          "--query_path", P["query.fbin"], "--gt_path", P["gt.ibin"], "--projection_index_save_path", P["roar.index"],
          "--L_pq", "30", "--k", "10", "--evaluation_save_path", P["eval.csv"]]
for extra in ([], ["--per_query", "-T", "1"], ["--per_query", "-T", "16"], ["--per_query", "-T", "64"], ["--per_query", "-T", "256"]):
    subprocess.check_call(common + extra, stdout=subprocess.DEVNULL)
    print(" ".join(extra) or "batch", "->", open(P["eval.csv"]).read().strip(), flush=True)
