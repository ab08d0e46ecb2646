"""The reference's whole evaluation sweep (run_roargraph_search_test.sh:13, 56 values of L_pq, k = 10)
on the T2I-10M shape: one CSV row per L_pq in the reference's column order
  L_pq,qps,avg_cmps,mean_latency_ms,recall,avg_hops
for the GPU (10 000 queries per launch, kernel time) -- avg_cmps is the REFERENCE algorithm's count (EXACT visited
mode, untimed second run), kernel_cmps what the timed kernel scored (LDS mode re-scores nodes its tag table forgot),
roofline_frac = reference-algorithm bytes / kernel time / 8 TB/s -- and for a subset of L_pq the CPU reference-kind
baseline (16 threads = the box's CPU quota) on the first --cpu-queries queries together with an exact
comparison of the ids / distances of both sides.  Test infrastructure (runs the oracle)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
import bench
from mysteryann_amd import index as mi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--cpu-queries", type=int, default=2000)
ap.add_argument("--out", default="gpurun_out/sweep_t2i10m.csv")
ap.add_argument("--no-cpu", action="store_true")
ap.add_argument("--builder", default="reference")
ap.add_argument("--graph-cache", default="/tmp/roar_graph.index")
args = ap.parse_args()
n, d, nq, k = args.n, 200, 10_000, 10
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "scripts"))
from _graph import t2i_index
idx, base, queries = t2i_index(n, nq, args.builder, 0, args.graph_cache)
gt = idx.ExactKnnDevice(queries, k, want_dists=False)[0].to(torch.int64)
impl = "ref" if oracle.have_ref() else "port"
threads = min(oracle.max_threads(impl), bench.effective_cpus())
hb = base.cpu().numpy(); ep, off, nbrs = idx.GetProjectionGraph(); hq = queries.cpu().numpy()
cpu_L = {10, 20, 50, 60, 100, 200, 500, 1000, 2000}
rows = ["L_pq,qps,avg_cmps,mean_latency_ms,recall,avg_hops,visited_mode,kernel_cmps,screen,vis_entries,hub_bitmap,grid,rows_8bit,rows_f32,roofline_frac,"
        "reference_bytes_rate_over_peak,lds_exact_ms,lds_exact_cmps,cpu_qps,cpu_threads,gpu_cpu_ids_identical"]
tblock = torch.zeros((nq, 4), dtype=torch.int32, device="cuda")
srow = idx.info("screen_row_bytes")
ap_cpu = not args.no_cpu
for L in bench.L_LIST:
    for _ in range(2): out = idx.SearchRoarGraphBatchDevice(queries, k, L)
    torch.cuda.synchronize()
    ms = idx.info("last_kernel_ns") / 1e6
    mode, ve, grid, hubf = idx.info("last_visited_mode"), idx.info("last_vis_entries"), idx.info("last_grid"), idx.info("last_hub_bitmap")
    forgetful, screen = idx.info("last_visited_forgetful"), idx.info("last_screen")
    idx.SetTrafficCounters(tblock)  # the same plan once more under the counting instantiation
    idx.SearchRoarGraphBatchDevice(queries, k, L); torch.cuda.synchronize()
    idx.SetTrafficCounters(None)
    tb = tblock.to(torch.float64).sum(dim=0)
    impl_bytes = bench.implemented_bytes(tb[0].item(), tb[1].item(), tb[2].item(), nq, 200, srow, k)
    idx.set_option("visited_mode", 1)  # the on-chip table in its exact form (full sets spill to HBM): cmps must be the reference's
    for _ in range(2): sp = idx.SearchRoarGraphBatchDevice(queries, k, L)
    torch.cuda.synchronize()
    sp_ms = idx.info("last_kernel_ns") / 1e6
    idx.set_option("visited_mode", 2)
    rec = bench.recall_at_k(torch, out["ids"], gt, k)
    idx.set_option("visited_mode", 0)  # the reference algorithm's counters
    ex = idx.SearchRoarGraphBatchDevice(queries, k, L); torch.cuda.synchronize()
    idx.set_option("visited_mode", 2)
    assert bool((ex["ids"] == out["ids"]).all()) and bool((ex["dists"] == out["dists"]).all()) and bool((ex["hops"] == out["hops"]).all())
    assert bool((sp["ids"] == ex["ids"]).all()) and bool((sp["cmps"] == ex["cmps"]).all()), "exact on-chip form: cmps must equal the reference's"
    alg = (800.0 * ex["cmps"].double().sum() + 4 * ex["scanned"].double().sum() + 8 * ex["hops"].double().sum()).item() + 880.0 * nq
    cpu_qps, same = "", ""
    if L in cpu_L and ap_cpu:
        m = args.cpu_queries if L <= 200 else max(200, args.cpu_queries // 5)
        oracle.search_batch(hb, off, nbrs, ep, hq[:64], k, L, 1, threads=threads, impl=impl)
        t0 = time.perf_counter(); o = oracle.search_batch(hb, off, nbrs, ep, hq[:m], k, L, 1, threads=threads, impl=impl); dt = time.perf_counter() - t0
        cpu_qps = f"{m / dt:.0f}"
        gi = out["ids"][:m].cpu().numpy().view(np.uint32); gd = out["dists"][:m].cpu().numpy()
        same = f"{float(((gi == o['ids']).all(axis=1) & (gd == o['dists']).all(axis=1)).mean()):.4f}"
    rows.append(f"{L},{nq / ms * 1e3:.0f},{float(ex['cmps'].float().mean()):.1f},{ms / nq:.6f},{rec:.4f},"
                f"{float(out['hops'].float().mean()):.2f},{('lds-forgetful' if forgetful else 'lds-exact') if mode == 1 else 'exact'},{float(out['cmps'].float().mean()):.1f},{screen},{ve},{hubf},{grid},"
                f"{tb[0].item() / nq:.1f},{tb[1].item() / nq:.1f},{impl_bytes / (ms * 1e-3) / 8e12:.3f},"
                f"{alg / (ms * 1e-3) / 8e12:.3f},{sp_ms:.3f},{float(sp['cmps'].float().mean()):.1f},{cpu_qps},{threads if cpu_qps else ''},{same}")
    print(rows[-1], flush=True)
os.makedirs(os.path.dirname(args.out), exist_ok=True)
open(args.out, "w").write("\n".join(rows) + "\n")
