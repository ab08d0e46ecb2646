"""Shared by the measurement scripts: the benchmark data set and index (bench.py's defaults), built once per box session
and cached as an index file so that several scripts of one gpurun call measure the same graph without rebuilding it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mysteryann_amd import index as mi, synth


def t2i_index(n=10_000_000, nq=10_000, builder="reference", ref_learn=0, cache="/tmp/r2_graph.index", say=print):
    d = 200
    n_learn = (ref_learn or n // 5) if builder == "reference" else n
    have = bool(cache) and os.path.exists(cache)
    base, queries, learn = synth.t2i_torch(n, nq, d, n_learn=0 if have else n_learn)
    idx = mi.IndexBipartite(d, n, mi.INNER_PRODUCT, None)
    idx.SetBaseData(base)
    t0 = time.time()
    if have:
        idx.LoadProjectionGraph(cache)
    elif builder == "reference":
        lists, _ = idx.ExactKnnDevice(learn, 100, want_dists=False)
        p = mi.Parameters(); p.Set("M_sq", 100); p.Set("M_pjbp", 35); p.Set("L_pjpq", 500)
        idx.BuildRoarGraphDevice(lists, p)
        del lists
    else:
        idx.BuildBootstrapGraph(learn=learn, max_degree=384, row_keep=128)
    torch.cuda.synchronize()
    if cache and not have:
        idx.SaveProjectionGraph(cache)
    say(f"# T2I shape n={n}: graph {'loaded from ' + cache if have else 'built (' + builder + ')'} in {time.time() - t0:.1f}s, avg degree "
        f"{idx.info('nedges') / n:.1f}, max {idx.info('max_degree')}, stride {idx.info('adj_stride')}, hash {idx.info('graph_hash') & (2 ** 64 - 1):016x}")
    return idx, base, queries
