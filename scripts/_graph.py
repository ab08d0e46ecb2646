"""Shared by the measurement scripts: the benchmark data set and index (bench.py's defaults), built once per box session
and cached as an index file so that several scripts of one gpurun call measure the same graph without rebuilding it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mysteryann_amd import index as mi, synth


def t2i_index(n=10_000_000, nq=10_000, builder="reference", ref_learn=0, cache="/tmp/roar_graph.index", say=print, stage2=-1, stage2_L=600):
    """bench.py's default index (exact lists of as many training queries as base points, bench.build_reference_graph) unless told
    otherwise (ref_learn < n: round 3's two-stage lists)."""
    d = 200
    n_exact = ref_learn or n
    n_learn = max(n_exact, n if stage2 < 0 else stage2) if builder == "reference" else n
    have = bool(cache) and os.path.exists(cache)
    base, queries, learn = synth.t2i_torch(n, nq, d, n_learn=0 if have else n_learn)
    idx = mi.IndexBipartite(d, n, mi.INNER_PRODUCT, None)
    idx.SetBaseData(base)
    t0 = time.time()
    if have:
        idx.LoadProjectionGraph(cache)
    elif builder == "reference":
        import bench
        bench.build_reference_graph(torch, mi, idx, learn, n_exact, stage2_L, lambda m: say("# " + m))
    else:
        idx.BuildBootstrapGraph(learn=learn, max_degree=384, row_keep=128)
    torch.cuda.synchronize()
    if cache and not have:
        idx.SaveProjectionGraph(cache)
    say(f"# T2I shape n={n}: graph {'loaded from ' + cache if have else 'built (' + builder + ')'} in {time.time() - t0:.1f}s, avg degree "
        f"{idx.info('nedges') / n:.1f}, max {idx.info('max_degree')}, stride {idx.info('adj_stride')}, hash {idx.info('graph_hash') & (2 ** 64 - 1):016x}")
    return idx, base, queries
