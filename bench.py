#!/usr/bin/env python3
"""bench.py -- QPS @ recall@10 >= 0.9 on the Text-to-Image-10M shape (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N > 1: either launched by torch.distributed.run, one rank per
                                                          GPU, or typed as is -- it then starts those ranks itself as a
                                                          child process, before anything here touches the GPU)

One "step" = one pass of the hot path (the batched RoarGraph beam search, one kernel launch per GPU) over ONE
10 000-query OOD batch, inputs already resident in HBM.  Workload (config.workload "t2i-10M-ip"): synthetic
base 10 M x 200-d f32 + out-of-distribution queries (mysteryann_amd/synth.py), inner-product
metric, k = 10, index built in HBM by the reference's own algorithm (BuildRoarGraph, canonical parameters, exact
learn->base lists of n / 5 training queries from the MFMA kNN generator) and replicated on every GPU.  Eight distinct query batches rotate through
the timed loop (a replayed batch would find its rows in the Infinity Cache); the replayed figure is reported
beside it.  Multi-GPU (BASELINE configs[3]): the batch is split by query, rank r searches rows
[r nq/N, (r+1) nq/N) with no collective -> "scaling": "strong", value = nq x steps / max-over-ranks time
(--scaling weak: every rank its own 10 000-query batch).

L_pq is the smallest value of the reference's sweep list (run_roargraph_search_test.sh:13) whose
recall@10 against exact brute-force ground truth is >= 0.9.

Printed by rank 0: ONE JSON line with the driver's contract fields plus
  roofline      HBM-gather roofline of the search kernel: algorithmic bytes (reference cmps / hops /
                scanned counters, SURVEY 8(d) formula) / kernel time measured with HIP events
  cpu_baseline  the reference's own distance / queue / visited-pool code (oracle/_ref, "reference")
                or the C restatement ("port") timed on this box's host cores on the same queries
  value_host_buffers  (N = 1) the same search through roar_search_batch with host buffers, the drop-in call
--sharded runs the index-sharded configuration instead (BASELINE configs[4]): one shard per rank, RCCL all-gather
of the per-shard top-k, HIP merge; same JSON contract.
Diagnostics go to stderr.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Launches of several steps overlap on several HIP streams (--in-flight): give the runtime as many hardware queues as streams
# (its default is four per device; a launch can only start when the launch before it in the same hardware queue has finished
# -- its slowest query -- so with four queues six streams of 1 250-query launches run at 0.270 ms per step, with eight at
# 0.240 ms; scripts/scale_probe.py).  Read by the HIP runtime when it starts, i.e. before torch is imported; inherited by the
# ranks this script starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

L_LIST = [10, 15, 20, 25, 30, 35, 40, 45, 50, 55, 60, 65, 70, 75, 80, 85, 90, 95, 100, 110, 120, 130, 140, 150, 160,
          170, 180, 190, 200, 220, 240, 260, 280, 300, 350, 400, 450, 500, 550, 600, 650, 700, 750, 800, 900, 1000,
          1100, 1200, 1300, 1400, 1500, 1600, 1700, 1800, 1900, 2000]  # run_roargraph_search_test.sh:13
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured streaming copy)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def brute_force_gt(torch, base, d, queries, k, chunk_rows=1 << 20, chunk_q=2048, l2=False):
    """Exact top-k by inner product (max dot) or L2, ids only.  Test infrastructure (torch matmul)."""
    nq = queries.shape[0]
    out = torch.empty((nq, k), dtype=torch.int64, device=queries.device)
    for q0 in range(0, nq, chunk_q):
        q = queries[q0:q0 + chunk_q, :d]
        best_v = torch.full((q.shape[0], k), -float("inf"), device=q.device)
        best_i = torch.zeros((q.shape[0], k), dtype=torch.int64, device=q.device)
        for r0 in range(0, base.shape[0], chunk_rows):
            blk = base[r0:r0 + chunk_rows, :d]
            s = q @ blk.T
            if l2:  # -(|x|^2 - 2 q.x): larger is nearer
                s = 2 * s - (blk * blk).sum(dim=1)[None, :]
            v, i = torch.topk(s, k, dim=1)
            v = torch.cat([best_v, v], 1)
            i = torch.cat([best_i, i + r0], 1)
            best_v, sel = torch.topk(v, k, dim=1)
            best_i = torch.gather(i, 1, sel)
        out[q0:q0 + chunk_q] = best_i
    return out


def build_reference_graph(torch, mi, idx, learn, n_exact, stage2_L, say=log):
    """The benchmark's index: the reference's BuildRoarGraph (canonical parameters M_sq = 100, M_pjbp = 35, L_pjpq = 500,
    run_roargraph_test.sh:9) on learn->base lists of ALL of `learn` (the reference's own T2I-10M run has as many training
    queries as base points).  Round 4: every list is EXACT (n_exact = all: the bf16-screened generator makes 10 M lists in ~75 s;
    rounds 1-3's f32 MFMA form needed 500 s).  With n_exact < len(learn) -- round 3's default, kept as an option -- only the
    first n_exact queries get exact lists; those make a first graph, the lists of the others are SEARCHED on it (k = 100,
    L_pq = stage2_L), and the final graph is built from all lists by the same BuildRoarGraph (T2I-10M, 2 M exact + 8 M
    searched at L_pq = 600: recall@10 at L_pq = 50 0.9011; 10 M exact lists: 0.9005).  -> description"""
    import time as _t

    n_learn = learn.shape[0]
    n_exact = min(n_exact, n_learn)
    t0 = _t.time()
    lists, _ = idx.ExactKnnDevice(learn[:n_exact], 100, want_dists=False)
    torch.cuda.synchronize()
    say(f"exact learn->base kNN ({n_exact} x 100, {'bf16 MFMA screen + exact f32 re-rank' if idx.info('last_knn_bf16') else 'f32 MFMA'}) "
        f"in {_t.time() - t0:.1f}s")
    p = mi.Parameters()
    p.Set("M_sq", 100), p.Set("M_pjbp", 35), p.Set("L_pjpq", 500)
    idx.BuildRoarGraphDevice(lists, p)
    if n_learn <= n_exact:
        return f"reference shape: exact lists of {n_exact} training queries, M_sq=100 M_pjbp=35 L_pjpq=500"
    t1 = _t.time()
    all_lists = torch.empty((n_learn, 100), dtype=torch.int32, device=learn.device)
    all_lists[:n_exact] = lists
    del lists
    for a in range(n_exact, n_learn, 50_000):
        b_ = min(n_learn, a + 50_000)
        all_lists[a:b_] = idx.SearchRoarGraphBatchDevice(learn[a:b_].contiguous(), 100, stage2_L)["ids"]
    torch.cuda.synchronize()
    say(f"searched lists of {n_learn - n_exact} more training queries (k=100, L_pq={stage2_L}) in {_t.time() - t1:.1f}s")
    idx.BuildRoarGraphDevice(all_lists, p)
    return (f"reference shape, two stages: exact lists of {n_exact} training queries -> graph -> lists of {n_learn - n_exact} more "
            f"searched on it (k=100, L_pq={stage2_L}) -> BuildRoarGraph on all {n_learn} lists, M_sq=100 M_pjbp=35 L_pjpq=500")


def implemented_bytes(rows_code, rows_f32, adj_words, queries, dim_padded, screen_row_bytes, k):
    """Bytes the IMPLEMENTED algorithm asks the memory side for (the counters are the kernel's own,
    roar_set_traffic_counters_device): every row scored on the 8-bit screening copy is one row of that copy, every row that reaches
    the exact pass one f32 row, every adjacency fetch its whole 64-word chunks (speculative fetches that were dropped included),
    + the query read and the result write.  SURVEY 8(d)'s formula with the reference's `4 D cmps` split into the two kinds of
    row this kernel reads; visited-set traffic is excluded there and here."""
    return (float(screen_row_bytes) * float(rows_code) + 4.0 * dim_padded * float(rows_f32) + 4.0 * float(adj_words)
            + (4.0 * dim_padded + 8.0 * k) * float(queries))


def share_of_8_block(nq, steps, wall_share, wall_share_serial, depth, cap, value_this_run, wall_stream=None, stream_info=None):
    """config.emulated_share_of_8: what the strong-scaling run at N = 8 asks of each rank -- nq / 8 queries per step -- timed on THIS GPU
    over the same K steps (wall seconds through one batch stream / with `depth` launches in flight / one at a time).  The eight ranks
    run side by side, so the projected whole-job rate is nq x K / that wall time; its ratio to this run's own value is the projected
    speed-up.  The regime is the better of {batch stream, launches in flight}, named in `regime`, as for `value` itself."""
    use_stream = wall_stream is not None and wall_stream < wall_share
    best = wall_stream if use_stream else wall_share
    return {"queries_per_step": nq // 8, "regime": "one batch stream (roar_search_stream_*)" if use_stream else f"{depth} launches in flight",
            "ms_per_step": best * 1e3 / steps,
            "projected_value_8_gpus": nq * steps / best,
            "projected_speedup_over_this_run": (nq * steps / best) / value_this_run,
            "ms_per_step_batch_stream": None if wall_stream is None else wall_stream * 1e3 / steps,
            "ms_per_step_launches_in_flight": wall_share * 1e3 / steps, "launches_in_flight": depth, "workgroups_per_launch_cap": cap,
            "ms_per_step_one_launch_at_a_time": wall_share_serial * 1e3 / steps,
            "batch_stream": stream_info,
            "note": "a projection from ONE GPU: the first nq / 8 rows of every batch on this GPU stand for one of eight ranks"}


def kernel_trace_span(csv_path, steps, name="roar_search_v2_kernel"):
    """(span ms, mean duration ms) of the last `steps` launches of the search kernel in a rocprofv3 kernel_trace.csv: span = last
    end - first start (launches that overlap on several streams: span / steps is the effective time per launch)."""
    import csv

    rows = [x for x in csv.DictReader(open(csv_path)) if name in x["Kernel_Name"] and "build" not in x["Kernel_Name"]]
    if len(rows) < steps:
        raise RuntimeError(f"{len(rows)} traced launches of the search kernel in {csv_path}, expected at least {steps}")
    rows.sort(key=lambda x: int(x["Start_Timestamp"]))
    rows = rows[-steps:]
    t0 = min(int(x["Start_Timestamp"]) for x in rows)
    t1 = max(int(x["End_Timestamp"]) for x in rows)
    mean = sum(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in rows) / steps
    return (t1 - t0) * 1e-6, mean * 1e-6


def recall_at_k(torch, ids, gt, k):
    hit = (gt[:, :k, None] == ids[:, None, :k].to(torch.int64)).any(dim=2).sum().item()
    return hit / float(k * ids.shape[0])


def free_port():
    """A TCP port nobody listens on right now (for the rendezvous of a self-started N > 1 run)."""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def launcher_command(argv, gpus, port):
    """Command line that runs this script as `gpus` ranks of one node, one per GPU, over RCCL -- what the driver types
    for N > 1.  argv = this process's own arguments (sys.argv[1:]), passed through unchanged."""
    # (the launcher's parser would read a bare `--n` behind the script as an abbreviation of its own --nnodes / --nproc-per-node /
    # ...: spell it out)
    argv = ["--base-points" if a == "--n" else ("--base-points=" + a[4:] if a.startswith("--n=") else a) for a in argv]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv


def self_launch(args, argv):
    """`python bench.py --gpus N` typed as a plain command with N > 1: start the N ranks as a CHILD process and hand its
    exit code back.  This runs before torch is imported, i.e. before anything in this process has touched the GPU (a
    process that has must never be replaced by exec; a child is always fine).  -> exit code, or None to go on here."""
    if args.gpus <= 1 or "RANK" in os.environ or args.pmc_child:
        return None
    import subprocess

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = launcher_command(argv, args.gpus, free_port())
    log("[bench] --gpus %d without a launcher: starting %s" % (args.gpus, " ".join(cmd)))
    return subprocess.call(cmd, env=env)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="t2i", choices=["t2i", "webvid", "sift", "t2i-hard"],
                    help="t2i = the BASELINE metric's configuration (default); webvid / sift = the other configs' shapes; t2i-hard = the "
                         "t2i shape on an isotropic base set (latent = D, optionally a common offset: --hard-offset), the data the "
                         "8-bit screen was NOT built for (mysteryann_amd/synth.py)")
    ap.add_argument("--hard-offset", type=float, default=0.0,
                    help="t2i-hard: common component of every base row, in units of the isotropic part's norm (a narrow cone)")
    ap.add_argument("--n", "--base-points", dest="n", type=int, default=0, help="base points (default: 10 M t2i, 2.5 M webvid, 1 M sift)")
    ap.add_argument("--nq", type=int, default=10_000, help="queries in one batch (the reference's query file: 10 000)")
    ap.add_argument("--batches", type=int, default=8,
                    help="distinct query batches rotated through the timed loop (same distribution, own draws), so that a "
                         "step never finds the rows of the step before it in the Infinity Cache")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="launches in flight in the timed loop (step s on stream s mod IN_FLIGHT; each step its own batch and result "
                         "buffers).  1 = one launch after the other; 0 = 6 (a launch ends with its slowest query: launches have to "
                         "overlap to keep the chip full).  The kernel's own duration (roofline) is always measured one launch at a time")
    ap.add_argument("--launch-grid", type=int, default=-1,
                    help="workgroups per launch while launches overlap (-1 = 1 024, 640 for a rank's share under 8 000 queries; 0 = no cap)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = ONE nq-query batch per step, rank r searches rows [r nq/N, (r+1) nq/N) of it "
                         "(BASELINE configs[3]); weak = every rank searches its own nq-query batch")
    ap.add_argument("--dim", type=int, default=0)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--L", type=int, default=0, help="0 = smallest L_pq of the reference sweep with recall >= target")
    ap.add_argument("--target-recall", type=float, default=0.9)
    ap.add_argument("--visited", default="auto", choices=["auto", "lds", "exact", "forgetful"],
                    help="visited-set form: auto (the fastest per L_pq: the forgetful on-chip table, HBM tags for the longest pools), "
                         "lds (on-chip table, exact: full sets spill to HBM), forgetful (on-chip table, a full set overwrites), "
                         "exact (HBM tags only)")
    ap.add_argument("--build-M", type=int, default=16)
    ap.add_argument("--build-basecap", type=int, default=64)
    ap.add_argument("--build-maxdeg", type=int, default=384,
                    help="room in which a row of the built graph collects proposals (hub rows fill any room they get)")
    ap.add_argument("--build-keep", type=int, default=128,
                    help="> 0: after every build phase a row longer than KEEP keeps its KEEP nearest neighbours.  T2I-10M, "
                         "recall 0.9: first come first kept, room 128 / 200 -> L_pq 55 / 30 at 5.83 / 5.52 ms per 10 k queries; "
                         "room 384, KEEP 128 -> L_pq 30 at 3.92 ms (KEEP 184, room 512: L_pq 20 at 3.96 ms)")
    ap.add_argument("--build-L", type=int, default=48)
    ap.add_argument("--build-Msq", type=int, default=48)
    ap.add_argument("--build-Mpj", type=int, default=24)
    ap.add_argument("--build-Llearn", type=int, default=128)
    ap.add_argument("--build-rounds", type=int, default=2)
    ap.add_argument("--build-supply", type=int, default=12)
    ap.add_argument("--build-supplyL", type=int, default=128)
    ap.add_argument("--build-Llast", type=int, default=256)
    ap.add_argument("--build-C1", type=int, default=48, help="phase-1 candidates to prune from (0 = unpruned)")
    ap.add_argument("--build-C3", type=int, default=48, help="supply candidates to prune from (0 = unpruned)")
    ap.add_argument("--build-projmode", type=int, default=0)
    ap.add_argument("--build-supplymode", type=int, default=0)
    ap.add_argument("--build-trim", type=int, nargs=2, default=[8, 0], metavar=("MIN_PROJ", "KEEP_BASE"))
    ap.add_argument("--builder", default="reference", choices=["reference", "bootstrap"],
                    help="reference (default) = roar_build_roargraph: the reference's BuildRoarGraph / LinkProjection on exact "
                         "learn->base lists with its canonical parameters M_sq 100 / M_pjbp 35 / L_pjpq 500 "
                         "(run_roargraph_test.sh:9); bootstrap = roar_build_graph (no exact lists needed)")
    ap.add_argument("--ref-learn", type=int, default=0,
                    help="--builder reference: training queries whose exact top-100 lists are computed (0 = all of them for the t2i "
                         "workload -- the reference's own input; n / 5 for the other shapes).  Fewer + --ref-learn2: round 3's two-stage lists")
    ap.add_argument("--ref-learn2", type=int, default=-1,
                    help="--builder reference, second stage: this many training queries in all (-1 = default: as many as base points, the "
                         "reference's own T2I-10M setting, for the t2i workload; 0 for the others = no second stage).  The first --ref-learn of them get exact lists (MFMA brute force) and make a first "
                         "graph; the lists of the others are SEARCHED on that graph (k = 100, L_pq = --ref-learn2-L: 10 M searches take "
                         "seconds where 10 M exact lists take 500 s), and the final graph is built from all lists by the same "
                         "BuildRoarGraph with the same canonical parameters")
    ap.add_argument("--ref-learn2-L", type=int, default=600)
    ap.add_argument("--graph-cache", default="",
                    help="index file: loaded if it exists (LoadProjectionGraph), else written after the build -- lets the "
                         "profiling passes of one box session (scripts/profile_r2.sh) skip the four-minute build; the graph "
                         "hash in the JSON says it is the same graph")
    ap.add_argument("--max-L", type=int, default=2000)
    ap.add_argument("--exact-knn", type=int, default=0,
                    help="exact learn->base kNN lists (MFMA brute force) for the first N training queries")
    ap.add_argument("--torch-gt", action="store_true", help="ground truth by torch matmul+topk instead of the MFMA generator")
    ap.add_argument("--build-phase1-ip", action="store_true", help="phase 1 by the index metric instead of L2")
    ap.add_argument("--learn", type=int, default=-1,
                    help="training queries for the projection pass (-1 = n, the reference's T2I-10M setting: as many "
                         "training queries as base points)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the host-buffer (drop-in) timing")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE) that measure roofline.traffic")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # the profiled child of measure_traffic()
    ap.add_argument("--plan", default="", help=argparse.SUPPRESS)  # --pmc-child: the launch plan the parent's calibration picked (screen,mode,tags)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-share-probe", action="store_true",
                    help="N = 1: skip the emulated share of 8 (the timed loop once more over the first nq / 8 rows of every batch)")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="default t2i run: skip the compact runs of the other BASELINE shapes (webvid 2.5 M x 512, sift 1 M x 128) that are "
                         "appended to the line as `other_workloads`")
    ap.add_argument("--no-stream", action="store_true", help="skip the batch-stream regime (roar_search_stream_*): launches in flight only")
    ap.add_argument("--no-screen", action="store_true", help="switch the 8-bit screening pass off (option screen = 1)")
    ap.add_argument("--force-screen", action="store_true", help="8-bit screening pass on wherever there is a kernel (option screen = 2) instead of measured")
    ap.add_argument("--vis-entries", type=int, default=0, help="LDS visited-table entries (0 = auto)")
    ap.add_argument("--tune", action="store_true", help="time the chosen L_pq over visited-table sizes (stderr)")
    ap.add_argument("--sweep", action="store_true", help="print the whole L_pq curve (CSV columns of the reference)")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--sharded", action="store_true",
                    help="BASELINE configs[4]: the base set split over the ranks (12.5 M points each by default, own graph "
                         "per shard), every rank searches all queries, RCCL all-gather of the per-shard top-k, HIP merge")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    rc = self_launch(args, sys.argv[1:])
    if rc is not None:
        raise SystemExit(rc)

    import numpy as np
    import torch

    shape = {"t2i": (10_000_000, 200), "webvid": (2_500_000, 512), "sift": (1_000_000, 128), "t2i-hard": (10_000_000, 200)}[args.workload]
    args.n_explicit = args.n
    args.n = args.n or shape[0]
    args.dim = args.dim or shape[1]
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path for the product")
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # launched by torch.distributed.run
    if args.gpus != world:  # (a plain `python bench.py --gpus N` never gets here: self_launch() started the ranks)
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: the launcher's --nproc-per-node and --gpus must agree")
    if torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but this node shows {torch.cuda.device_count()} GPU(s)")
    torch.cuda.set_device(local)
    if use_dist:
        import torch.distributed as dist

        # RCCL prints a version banner on STDOUT when its first communicator comes up; stdout is for the one JSON line, so the file
        # descriptor points at stderr until that has happened
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    dev = torch.device("cuda", local)
    from mysteryann_amd import index as mi
    from mysteryann_amd import parallel, synth

    ranks_seen = parallel.ranks_seen(dev)  # a one-element all-reduce of ones over RCCL: how many ranks the collective really spans

    if args.sharded:
        run_sharded(args, np, torch, mi, synth, dev, rank, world, use_dist)
        if use_dist:
            dist.destroy_process_group()
        return

    strong = args.scaling == "strong"
    B = max(1, args.batches)
    t0 = time.time()
    n_learn = args.learn if args.learn >= 0 else args.n
    n_exact = 0
    if args.builder == "reference":
        # exact lists of ALL training queries for the t2i workload (as many as base points: the reference's own T2I-10M setting,
        # run_roargraph_test.sh:5-9; 75 s with the bf16-screened generator), of n / 5 for the other shapes
        n_exact = args.ref_learn or (args.n if args.workload == "t2i" else args.n // 5)
        stage2 = args.ref_learn2 if args.ref_learn2 >= 0 else (args.n if args.workload == "t2i" else 0)
        n_learn = max(n_exact, stage2)
    gen = {"t2i": synth.t2i_torch, "webvid": synth.webvid_torch, "sift": synth.sift_torch,
           "t2i-hard": lambda *a, **kw: synth.t2i_hard_torch(*a, offset=args.hard_offset, **kw)}[args.workload]
    # B batches of nq queries in one draw; strong scaling: every rank draws the SAME block and takes its rows of each batch
    base, qall, learn = gen(args.n, args.nq * B, args.dim, seed=args.seed, n_learn=n_learn, device=dev,
                            query_seed=args.seed + 2 + (0 if strong else 1000 * rank))
    lo, hi = parallel.shard_range(args.nq, world, rank) if strong else (0, args.nq)
    batches = [qall[b * args.nq + lo:b * args.nq + hi].contiguous() for b in range(B)]
    mine = hi - lo
    total_q = args.nq if strong else args.nq * world  # queries all ranks process per step
    metric = mi.L2 if args.workload == "sift" else mi.INNER_PRODUCT
    torch.cuda.synchronize()
    log(f"[rank {rank}] synthetic {args.workload}-shape data: base {tuple(base.shape)}, {B} query batches of {args.nq} "
        f"(this rank: rows {lo}..{hi} of each), learn {n_learn} in {time.time() - t0:.1f}s")

    idx = mi.IndexBipartite(args.dim, args.n, metric, None, device=local)
    idx.SetBaseData(base)
    learn_knn = None
    t0 = time.time()
    cached = bool(args.graph_cache) and os.path.exists(args.graph_cache)
    if cached:
        idx.LoadProjectionGraph(args.graph_cache)
        index_desc = f"loaded from {os.path.basename(args.graph_cache)} (written by an earlier run of this command on this box)"
    elif args.builder == "reference":
        index_desc = build_reference_graph(torch, mi, idx, learn, n_exact, args.ref_learn2_L, lambda m_: log(f"[rank {rank}] {m_}"))
    else:
        if args.exact_knn and n_learn:
            ne = min(args.exact_knn, n_learn)
            learn_knn, _ = idx.ExactKnnDevice(learn[:ne], args.build_Msq, want_dists=False)
            torch.cuda.synchronize()
            log(f"[rank {rank}] exact learn->base kNN ({ne} x {args.build_Msq}, MFMA generator) in {time.time() - t0:.1f}s")
        idx.BuildBootstrapGraph(M=args.build_M, base_cap=args.build_basecap, max_degree=args.build_maxdeg,
                                L_build=args.build_L, learn=learn if n_learn else None, M_sq=args.build_Msq,
                                M_pj=args.build_Mpj, L_learn=args.build_Llearn, rounds=args.build_rounds,
                                phase1_l2=not args.build_phase1_ip, learn_knn=learn_knn, supply_k=args.build_supply,
                                supply_L=args.build_supplyL, L_learn_last=args.build_Llast,
                                trim_min_proj=args.build_trim[0], trim_keep_base=args.build_trim[1],
                                phase1_cand=args.build_C1, supply_cand=args.build_C3, proj_mode=args.build_projmode,
                                supply_mode=args.build_supplymode, row_keep=args.build_keep)
        index_desc = (f"skeleton M={args.build_M}/{args.build_C1} cap={args.build_basecap}/{args.build_maxdeg}/{args.build_keep} "
                      f"L={args.build_L}; projection learn={n_learn} Msq={args.build_Msq} Mpj={args.build_Mpj} "
                      f"L={args.build_Llearn} x{args.build_rounds}; supply {args.build_supply}/{args.build_C3} "
                      f"L={args.build_supplyL}; trim {args.build_trim[0]}/{args.build_trim[1]}")
    torch.cuda.synchronize()
    build_s = time.time() - t0
    if args.graph_cache and not cached and rank == 0:
        idx.SaveProjectionGraph(args.graph_cache)
    graph_hash = idx.info("graph_hash") & (2 ** 64 - 1)
    log(f"[rank {rank}] graph ({args.builder}): {build_s:.1f}s, avg degree {idx.info('nedges') / args.n:.1f}, "
        f"max degree {idx.info('max_degree')}, ep {idx.info('ep')}, adjacency stride {idx.info('adj_stride')} words, "
        f"hash {graph_hash:016x}")
    del learn

    if args.pmc_child:  # profiled child of measure_traffic(): the timed loop's launches and nothing else
        if not cached or not args.L:
            raise SystemExit("--pmc-child needs --graph-cache (an existing file) and --L")
        idx.set_option("visited_mode", {"exact": 0, "lds": 1, "auto": 2, "forgetful": 3}[args.visited])
        idx.set_option("screen", 1 if args.no_screen else 2 if args.force_screen else 0)
        if args.vis_entries:
            idx.set_option("vis_entries", args.vis_entries)
        if args.plan:  # the plan the parent measured and timed: the same launches here, whatever this process would have measured
            scr_, mode_, tags_ = (int(v_) for v_ in args.plan.split(","))
            idx.set_option("screen", 2 if scr_ else 1)
            idx.set_option("visited_mode", mode_)
            idx.set_option("vis_entries", tags_)
        # the timed loop's launches in the regime asked for: one after the other, or --in-flight launches overlapping on as many
        # streams with the workgroup cap of the timed loop (a --pmc pass serialises dispatches: what it measures is the capped
        # launch's traffic; a --kernel-trace pass lets them overlap and records every launch's start and end)
        depth_c = max(1, args.in_flight)
        streams_c = [torch.cuda.current_stream(dev)] if depth_c == 1 else [torch.cuda.Stream(dev) for _ in range(depth_c)]
        if depth_c > 1 and args.launch_grid > 0:
            idx.set_option("max_grid", args.launch_grid)
        for st_ in streams_c:
            idx.SearchRoarGraphBatchDevice(batches[0], args.k, args.L, stream=st_)
        torch.cuda.synchronize()
        for s_ in range(args.warmup + args.steps):
            idx.SearchRoarGraphBatchDevice(batches[s_ % B] if s_ >= args.warmup else batches[0], args.k, args.L,
                                           stream=streams_c[s_ % depth_c])
        torch.cuda.synchronize()
        return

    t0 = time.time()
    gts = []
    for b in range(B):
        if args.torch_gt:
            gts.append(brute_force_gt(torch, base, args.dim, batches[b], args.k, l2=(metric == mi.L2)))
        else:  # the repo's own exact-kNN generator (f32 MFMA), SURVEY 8(f) #2
            gts.append(idx.ExactKnnDevice(batches[b], args.k, want_dists=False)[0].to(torch.int64))
    torch.cuda.synchronize()
    log(f"[rank {rank}] brute-force ground truth for {B} x {mine} queries in {time.time() - t0:.1f}s "
        f"({'torch' if args.torch_gt else 'roar_exact_knn_device'})")

    VIS = {"exact": 0, "lds": 1, "auto": 2, "forgetful": 3}
    idx.set_option("screen", 1 if args.no_screen else 2 if args.force_screen else 0)
    idx.set_option("visited_mode", VIS[args.visited])
    if args.vis_entries:
        idx.set_option("vis_entries", args.vis_entries)

    def run(L, b=0, out=None, stream=None):
        return idx.SearchRoarGraphBatchDevice(batches[b], args.k, L, out=out, stream=stream)

    def global_recall(ids, gt):
        """recall@k over the whole batch: hits of this rank's rows, summed over the ranks"""
        hits = recall_at_k(torch, ids, gt, args.k) * ids.shape[0]
        if use_dist:
            t = torch.tensor([hits], device=dev, dtype=torch.float64)
            dist.all_reduce(t)
            hits = t.item()
        return hits / total_q

    # choose L_pq: the smallest value of the reference's list whose recall, averaged over ALL the batches that rotate through the
    # timed loop, reaches the target (the reported recall is that average, so it cannot come out under the target)
    curve = []
    chosen = args.L
    if chosen == 0 or args.sweep:
        for L in L_LIST:
            if L < args.k or L > args.max_L:
                continue
            r = 0.0
            for b in range(B - 1, -1, -1):  # (batch 0 last: its launch is the one timed below)
                o = run(L, b)
                torch.cuda.synchronize()
                r += global_recall(o["ids"], gts[b]) / B
            ns = idx.info("last_kernel_ns")
            curve.append((L, mine / (ns * 1e-9), float(o["cmps"].float().mean()), ns * 1e-6 / max(mine, 1), r,
                          float(o["hops"].float().mean())))
            log(f"[rank {rank}] L_pq={L:5d} recall@{args.k}={r:.4f} kernel={ns / 1e6:.3f} ms "
                f"cmps={curve[-1][2]:.0f} hops={curve[-1][5]:.1f}")
            if chosen == 0 and r >= args.target_recall:
                chosen = L
                if not args.sweep:
                    break
        if chosen == 0:
            chosen = max(L for L in L_LIST if L <= args.max_L)
            log(f"[rank {rank}] WARNING: recall target {args.target_recall} not reached, using L_pq={chosen}")
    if args.sweep and rank == 0:
        log("L_pq,qps,avg_cmps,mean_latency_ms,recall,avg_hops")
        for row in curve:
            log(",".join(f"{v:.6g}" for v in row))

    if args.tune:
        for ve in (1024, 2048, 4096, 8192, 16384, 32768):
            idx.set_option("vis_entries", ve)
            for _ in range(2):
                o = run(chosen)
            torch.cuda.synchronize()
            log(f"[tune] L_pq={chosen} vis_entries={ve}: kernel {idx.info('last_kernel_ns') / 1e6:.3f} ms, "
                f"cmps {float(o['cmps'].float().mean()):.0f}, grid {idx.info('last_grid')}, lds {idx.info('last_lds_bytes')}")
        idx.set_option("vis_entries", args.vis_entries)

    # per batch: recall at the chosen L_pq; the reference algorithm's counters (EXACT visited mode, untimed) for the
    # reference-priced bytes; and what the timed launch itself asks the memory side for (the counting instantiation of the same
    # kernel on the same launch plan, untimed: rows on the 8-bit copy, rows on the f32 rows, adjacency words)
    dpad = idx.info("dim_padded")
    srow = idx.info("screen_row_bytes")
    recalls, alg_bytes, impl_bytes, same_ids = [], [], [], True
    ref_cmps = ref_hops = ref_scan = lds_cmps = 0.0
    rows_code = rows_f32 = adj_words = rows_passed = 0.0
    counters_equal = True
    outs = []
    tblock = torch.zeros((mine, 4), dtype=torch.int32, device=dev)
    for b in range(B):
        o = run(chosen, b)
        torch.cuda.synchronize()
        recalls.append(global_recall(o["ids"], gts[b]))
        idx.SetTrafficCounters(tblock)
        oc = run(chosen, b)
        torch.cuda.synchronize()
        idx.SetTrafficCounters(None)
        tb = tblock.to(torch.float64).sum(dim=0)
        counters_equal = counters_equal and all(bool((oc[key_] == o[key_]).all().item()) for key_ in ("ids", "dists", "cmps", "hops"))
        impl_bytes.append(implemented_bytes(tb[0].item(), tb[1].item(), tb[2].item(), mine, dpad, srow, args.k))
        rows_code += tb[0].item() / B / mine
        rows_f32 += tb[1].item() / B / mine
        adj_words += tb[2].item() / B / mine
        # rows the screen let through = rows on f32 - entry point - neighbours that never saw the screen
        rows_passed += (tb[1].item() - mine - (oc["cmps"].double().sum().item() - tb[0].item())) / B / mine
        idx.set_option("visited_mode", 0)
        ex = run(chosen, b)
        torch.cuda.synchronize()
        idx.set_option("visited_mode", VIS[args.visited])
        alg_bytes.append((4 * dpad * ex["cmps"].double().sum() + 4 * ex["scanned"].double().sum()
                          + 8 * ex["hops"].double().sum() + (4 * dpad + 8 * args.k) * mine).item())
        same_ids = same_ids and bool((ex["ids"] == o["ids"]).all().item() and (ex["dists"] == o["dists"]).all().item())
        if b == 0:
            ex0_cmps = ex["cmps"].clone()
        ref_cmps += ex["cmps"].double().mean().item() / B
        ref_hops += ex["hops"].double().mean().item() / B
        ref_scan += ex["scanned"].double().mean().item() / B
        lds_cmps += o["cmps"].double().mean().item() / B
        outs.append(o)
    rec = float(np.mean(recalls))
    screen_pass_rate = (rows_passed / rows_code) if rows_code > 0 else None

    main_stream = torch.cuda.current_stream(dev)
    depth_wanted = args.in_flight if args.in_flight > 0 else 6
    side = [torch.cuda.Stream(dev) for _ in range(max(1, min(depth_wanted, B)))]
    for st_ in side:  # a stream's hardware queue is made on its first use (milliseconds): not inside a timed loop
        run(chosen, 0, outs[0], st_)
    torch.cuda.synchronize()

    def step_buffers(rows):
        """one set of result arrays per step of a timed loop whose steps overlap (steps s and s + B search the same batch: they must
        not share their result arrays)"""
        return [dict(ids=torch.empty((rows, args.k), dtype=torch.int32, device=dev), dists=torch.empty((rows, args.k), dtype=torch.float32, device=dev),
                     cmps=torch.empty(rows, dtype=torch.int32, device=dev), hops=torch.empty(rows, dtype=torch.int32, device=dev),
                     scanned=torch.empty(rows, dtype=torch.int32, device=dev), short=torch.zeros(1, dtype=torch.int32, device=dev))
                for _ in range(max(args.steps, args.warmup, 1))]

    def poison(bufs):
        """Result arrays that a timed loop is about to write: filled with values no search produces (ids -1 in every column, NaN
        distances, hops -1), so that a step that wrote nothing -- a refused launch, a batch a stream never served -- cannot pass the
        comparison with the per-launch results on what an EARLIER loop left in the same arrays."""
        for o_ in bufs:
            o_["ids"].fill_(-1), o_["dists"].fill_(float("nan")), o_["hops"].fill_(-1), o_["cmps"].fill_(-1), o_["short"].zero_()
        torch.cuda.synchronize()

    def steps_equal_outs(bufs, what, keys=("ids", "dists", "hops")):
        """every timed step's results against the per-launch results of its batch (outs, made one launch at a time and checked against
        ground truth above); a refused launch (0xFFFFFFFF in its short count) or any difference FAILS the run: a number measured over
        steps that did not search is not a number."""
        for s_ in range(args.steps):
            if int(bufs[s_]["short"].item()) == -1:
                raise RuntimeError(f"bench: {what}: the launch of step {s_} was refused (dirty launch cell): it did not search")
            for key_ in keys:
                a_, b_ = bufs[s_][key_], outs[s_ % B][key_]
                same = bool((a_ == b_).all().item()) if key_ != "dists" else bool(torch.equal(a_, b_))
                if not same:
                    raise RuntimeError(f"bench: {what}: step {s_} wrote {key_} that differ from the per-launch results of batch {s_ % B}")
        return True

    def timed_stream(order, bufs):
        """The same K steps through ONE batch stream (roar_search_stream_*): every step's batch is submitted and waited for inside the
        timed region; workgroups are not bound to a step -- they take queries from whatever has been submitted and cross step
        boundaries, so a step's last queries run beside the next steps' first ones without a launch per step.  The stream is opened
        (buffers, plan) outside the region, as the HIP streams of the other loop are made outside it; its workgroups are started by
        the first submit inside.  -> (wall s max over ranks, GPU span ms from the device clock: first query taken .. last batch
        complete, info)"""
        st_ = idx.OpenSearchStream(args.k, chosen)
        try:
            tk = [st_.submit(batches[order(s_)], bufs[s_])[0] for s_ in range(args.warmup)]
            for t__ in tk:
                st_.wait(t__)
            t_end = time.time() + 2.0
            while st_.info("workgroups_alive") and time.time() < t_end:  # (the timed region starts from an idle stream)
                time.sleep(0.0005)
            torch.cuda.synchronize()
            poison(bufs)
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            launches0 = st_.info("launches")
            t_ = time.perf_counter()
            tk = [st_.submit(batches[order(s_)], bufs[s_])[0] for s_ in range(args.steps)]
            for t__ in tk:
                st_.wait(t__)
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            wall_ = time.perf_counter() - t_
            if use_dist:
                t = torch.tensor([wall_], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                wall_ = t.item()
            tt = [st_.times(t__) for t__ in tk]
            span_ms = (max(b_ for _, b_ in tt) - min(a_ for a_, _ in tt)) * 1e-6
            t0_ = min(a_ for a_, _ in tt)
            log("[stream timeline] us since the first query was taken, per step (first query taken, batch complete): "
                + " ".join(f"({(a_ - t0_) / 1e3:.0f},{(b_ - t0_) / 1e3:.0f})" for a_, b_ in tt) + f"; wall {wall_ * 1e6:.0f} us")
            info_ = {"kernel_launches_in_the_timed_region": st_.info("launches") - launches0, "resident_workgroups": st_.info("resident_capacity"),
                     "times_wait_had_to_start_workgroups": st_.info("heals"), "screen": bool(st_.info("screen")),
                     "vis_entries": st_.info("vis_entries"), "lds_bytes": st_.info("lds_bytes")}
        finally:
            st_.close()
        return wall_, span_ms, info_

    def timed(order, in_flight=1, bufs=None):
        """`steps` launches, batch of step s = order(s); barrier + synchronize on both sides; -> (wall s max over ranks, gpu ms).
        in_flight > 1: step s is launched on stream s mod in_flight, so that a step's last queries (a launch ends with its
        slowest query: 72 hops against 61 on average) run beside the next step's first ones; every step still searches its
        own whole batch into its own result buffers (`bufs`: one set per step) inside the timed region."""
        streams = [main_stream] if in_flight <= 1 else side[:in_flight]
        for s_ in range(args.warmup):
            run(chosen, order(s_), bufs[s_] if bufs else outs[order(s_)], streams[s_ % len(streams)])
        torch.cuda.synchronize()
        if bufs:
            poison(bufs)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t_ = time.perf_counter()
        e0.record(main_stream)
        for st_ in streams:
            if st_ is not main_stream:
                st_.wait_event(e0)
        for s_ in range(args.steps):
            run(chosen, order(s_), bufs[s_] if bufs else outs[order(s_)], streams[s_ % len(streams)])
        for st_ in streams:
            if st_ is not main_stream:
                main_stream.wait_stream(st_)
        e1.record(main_stream)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        wall_ = time.perf_counter() - t_
        if use_dist:
            t = torch.tensor([wall_], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall_ = t.item()
        return wall_, e0.elapsed_time(e1)  # events on the launch stream (torch's current stream is passed down)

    run(chosen, 0, outs[0])
    visited_desc = ("exact (u8 epoch tags in HBM)" if idx.info("last_visited_mode") == 0 else
                    "lds-forgetful (on-chip tag table, a full set overwrites a victim)" if idx.info("last_visited_forgetful") else
                    "lds-exact (on-chip tag table, full sets spill to u16 epoch tags in HBM)")
    # the same launch with the on-chip table in its EXACT form (cmps = the reference's): what exact counting costs here
    idx.set_option("visited_mode", 1)
    exact_form_ms, exact_form_cmps = 1e9, 0.0
    for _ in range(3):
        o_ = run(chosen, 0)
        torch.cuda.synchronize()
        exact_form_ms = min(exact_form_ms, idx.info("last_kernel_ns") * 1e-6)
        exact_form_cmps = o_["cmps"].double().mean().item()
        exact_form_equal = bool((o_["cmps"] == ex0_cmps).all().item())
    if idx.info("last_visited_forgetful"):  # no room in HBM for the spill lists: the launch ran the forgetful form, not what was asked for
        log(f"[rank {rank}] the exact on-chip form was degraded to the forgetful one (no memory for the spill lists): not reported")
        exact_form_ms = exact_form_cmps = exact_form_equal = None
    idx.set_option("visited_mode", VIS[args.visited])
    wall_replay, gpu_replay = timed(lambda s_: 0)           # the same batch every step (rows of the last step still cached)
    wall_serial, gpu_ms = timed(lambda s_: s_ % B)          # distinct batches in rotation, one launch after the other:
    if any(int(o_["short"].item()) == -1 for o_ in outs):   # (0xFFFFFFFF: a launch that found its cell dirty and did not search)
        raise RuntimeError("bench: one launch at a time: a launch was refused (dirty launch cell): it did not search")
    last_kernel_ms = idx.info("last_kernel_ns") * 1e-6      #   the kernel's own duration (roofline) comes from this loop
    kernel_ms = gpu_ms / args.steps
    depth = len(side)
    pipe_grid = 0
    if depth > 1:                                           # THE measurement: the same K steps, `depth` launches in flight
        # each launch of an overlapping stream of launches takes a share of the chip, not all of it: 1 024 workgroups (640 for
        # a rank's share of the batch at N > 1); six of them oversubscribe the 3 072 resident workgroups, which is what keeps
        # every slot busy while launches come and go (profiles/r03_scale_probe.txt: 10 000 queries 1.463 ms per step against
        # 1.522 with two uncapped launches; 5 000 0.750 / 0.817; 2 500 0.389 / 0.466; 1 250 0.212 / 0.307 ms)
        pipe_grid = args.launch_grid if args.launch_grid >= 0 else (1024 if mine >= 8000 else 640)
        idx.set_option("max_grid", pipe_grid)
        sbufs = step_buffers(mine)
        wall, gpu_pipe = timed(lambda s_: s_ % B, depth, sbufs)
        idx.set_option("max_grid", 0)
        steps_equal_outs(sbufs, f"{depth} launches in flight")
    else:
        wall, gpu_pipe = wall_serial, gpu_ms
        sbufs = step_buffers(mine)
    wall_launches, gpu_launches = wall, gpu_pipe
    # the same K steps through ONE batch stream (round 5): workgroups outlive a step.  `value` is the better of the two regimes and
    # config.value_regime names it; both are in the line.
    stream_info = None
    wall_stream = gpu_stream = None
    if not args.no_stream:
        wall_stream, gpu_stream, stream_info = timed_stream(lambda s_: s_ % B, sbufs)
        # (the step buffers were poisoned inside timed_stream, after its warm-up: what is compared here was written by the timed steps)
        stream_info["results_equal_per_launch_results"] = steps_equal_outs(sbufs, "batch stream")
        log(f"[rank {rank}] batch stream: {wall_stream * 1e3 / args.steps:.4f} ms per step (launches in flight: {wall_launches * 1e3 / args.steps:.4f}), "
            f"{stream_info}")
    use_stream = wall_stream is not None and wall_stream < wall_launches
    if use_stream:
        wall, gpu_pipe = wall_stream, gpu_stream
    ms_per_step = wall * 1e3 / args.steps
    qps = total_q * args.steps / wall
    # N > 1, strong scaling: the same K steps once more with every rank on WHOLE batches of its own (rank r starts the rotation at
    # batch r): what the N GPUs serve when each has its own stream of 10 000-query batches (per-GPU work fixed = weak scaling),
    # reported beside the strong figure.  Recall is not re-checked here (same index, same L_pq, same query distribution).
    weak_value = None
    if (world > 1 or os.environ.get("ROAR_BENCH_WEAK_SIDE")) and strong:  # (the variable: to exercise this loop on a one-GPU box)
        full = [qall[b * args.nq:(b + 1) * args.nq].contiguous() for b in range(B)]
        fouts = [idx.SearchRoarGraphBatchDevice(full[b], args.k, chosen) for b in range(B)]
        torch.cuda.synchronize()
        keep_b, keep_o = batches, outs
        batches, outs = full, fouts  # (run() and timed() read these names)
        idx.set_option("max_grid", 1024 if depth > 1 else 0)
        wall_w, _ = timed(lambda s_: (s_ + rank) % B, depth)
        idx.set_option("max_grid", 0)
        batches, outs = keep_b, keep_o
        weak_value = world * args.nq * args.steps / wall_w
    # ---- one GPU's share of the batch at N = 8, measured on this GPU (N = 1 only): the same K steps over the first nq / 8 rows of
    # every batch, the same depth, the cap of a rank's share -- what the strong-scaling run at N = 8 asks of each rank
    share8 = None
    if world == 1 and strong and args.nq >= 64 and not args.no_share_probe:
        sh = args.nq // 8
        keep_b, keep_o = batches, outs
        batches = [qall[b_ * args.nq:b_ * args.nq + sh].contiguous() for b_ in range(B)]
        outs = [idx.SearchRoarGraphBatchDevice(batches[b_], args.k, chosen) for b_ in range(B)]
        torch.cuda.synchronize()
        share_grid = args.launch_grid if args.launch_grid >= 0 else 640
        idx.set_option("max_grid", share_grid if depth > 1 else 0)
        shbufs = step_buffers(sh)
        wall_s8, gpu_s8 = timed(lambda s_: s_ % B, depth, shbufs)
        idx.set_option("max_grid", 0)
        steps_equal_outs(shbufs, "share of 8, launches in flight")
        wall_s8_1, _ = timed(lambda s_: s_ % B)
        if any(int(o_["short"].item()) == -1 for o_ in outs):
            raise RuntimeError("bench: share of 8, one launch at a time: a launch was refused (dirty launch cell)")
        wall_s8_stream = s8_info = None
        if not args.no_stream:
            wall_s8_stream, _, s8_info = timed_stream(lambda s_: s_ % B, shbufs)
            s8_info["results_equal_per_launch_results"] = steps_equal_outs(shbufs, "share of 8, batch stream")
        batches, outs = keep_b, keep_o
        share8 = share_of_8_block(args.nq, args.steps, wall_s8, wall_s8_1, depth, share_grid if depth > 1 else 0, qps, wall_s8_stream, s8_info)

    ref_bytes_per_launch = float(np.mean([alg_bytes[s_ % B] for s_ in range(args.steps)]))
    bytes_per_launch = float(np.mean([impl_bytes[s_ % B] for s_ in range(args.steps)]))
    # `value` is measured with `depth` launches in flight; the roofline line is stated in THAT regime: the effective duration of a
    # launch = GPU span of the K timed launches / K (HIP events on the launch streams: recorded on the main stream, which the
    # side streams wait for and which waits for them).  The one-launch-at-a-time figures stand beside it.
    eff_ms = gpu_pipe / args.steps
    achieved = bytes_per_launch / (eff_ms * 1e-3) / 1e9
    achieved_serial = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    regime = ("one batch stream (roar_search_stream_*), every step's batch submitted and waited for inside the timed region, workgroups cross "
              f"step boundaries: kernel_ms = device-clock span first query taken .. last batch complete / {args.steps}") if use_stream else \
        (f"{depth} launches in flight on {depth} streams, at most {pipe_grid} workgroups each: kernel_ms = GPU span of the {args.steps} "
         f"timed launches / {args.steps} (HIP events on the launch streams)") if depth > 1 else \
        "one launch at a time: kernel_ms = HIP events around the timed launches / steps"

    result = {
        "metric": "QPS @ recall@10=0.9, Text-to-Image-10M OOD" if (args.workload, args.n) == ("t2i", 10_000_000)
        else f"QPS @ recall@10={args.target_recall}, {args.workload} shape",
        "value": qps,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": args.scaling if world > 1 else "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "t2i-10M-ip" if (args.workload, args.n) == ("t2i", 10_000_000)
                   else f"{args.workload}-shape-n{args.n}" + (f"-offset{args.hard_offset:g}" if args.workload == "t2i-hard" else ""),
                   "n_base": args.n,
                   "dim": args.dim, "metric": "l2" if metric == mi.L2 else "ip", "n_queries_total": total_q,
                   "n_queries_per_gpu": mine, "k": args.k, "L_pq": chosen, "recall_at_10": rec,
                   "recall_at_10_min_over_batches": float(min(recalls)),
                   "value_regime": regime.split(":")[0],
                   "screen_pass_rate": screen_pass_rate,
                   "rccl_ranks_seen": ranks_seen,
                   "query_batches_rotated": B, "ms_per_step_same_batch_replayed": wall_replay * 1e3 / args.steps,
                   "launches_in_flight": depth, "workgroups_per_launch_cap_in_flight": pipe_grid,
                   "value_batch_stream": None if wall_stream is None else total_q * args.steps / wall_stream,
                   "value_launches_in_flight": total_q * args.steps / wall_launches,
                   "batch_stream": stream_info,
                   "value_with_a_whole_batch_per_rank_and_step_weak_scaling": weak_value, "ms_per_step_one_launch_at_a_time": wall_serial * 1e3 / args.steps,
                   "value_one_launch_at_a_time": total_q * args.steps / wall_serial,
                   "emulated_share_of_8": share8,
                   "parallelism": (f"index replicated on {world} GPU(s); one {args.nq}-query batch per step split by query, "
                                   f"rank r searches rows [r nq/N, (r+1) nq/N); no collective" if strong else
                                   f"index replicated on {world} GPU(s); every rank its own {args.nq}-query batch; no collective"),
                   "visited": visited_desc,
                   "screen": (f"8-bit codes of the base rows ({srow} B per row) scored first; a neighbour is dropped there only if its exact key is "
                              "certainly above the reject threshold; the others are scored on the f32 rows: ids, distances, hops, cmps unchanged")
                   if idx.info("last_screen") else "off",
                   "waves_per_query": idx.info("last_waves_per_query"), "register_pool": bool(idx.info("last_register_pool")),
                   "builder": args.builder, "index": index_desc, "graph_hash": f"{graph_hash:016x}",
                   "avg_degree": idx.info("nedges") / args.n, "adjacency_stride_words": idx.info("adj_stride"),
                   "build_seconds": build_s},
        # (the driver's record keeps the first keys of this object: what the line means comes first)
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "roar_search_v2_kernel", "kernel_ms": eff_ms, "regime": regime,
                     "requested_bytes_frac_of_peak": None, "l2_hit_share": None,
                     "frac_le_1": bool(achieved / HBM_PEAK_GBS <= 1.0),
                     "bytes_per_launch": bytes_per_launch,
                     "bytes": "IMPLEMENTED algorithm, from the kernel's own per-query counters: screen_row_bytes x rows scored on the 8-bit copy "
                              "+ 4 D_pad x rows scored on the f32 rows + 4 x adjacency words fetched (dropped speculative fetches included) "
                              "+ (4 D_pad + 8 k) per query; visited-set traffic excluded (SURVEY 8(d))",
                     "one_launch_at_a_time": {"kernel_ms": kernel_ms, "achieved": achieved_serial, "frac": achieved_serial / HBM_PEAK_GBS,
                                              "traffic": None, "requested_bytes_frac_of_peak": None,
                                              "last_launch_ms": last_kernel_ms,
                                              "kernel_ms_same_batch_replayed": gpu_replay / args.steps},
                     # the same launches priced in the REFERENCE algorithm's bytes (4 D per distance evaluation, reference
                     # cmps / hops / scanned): a rate the memory system never sees when the screen keeps most f32 rows from being read
                     "reference_bytes_speedup": ref_bytes_per_launch / (eff_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "reference_bytes_per_launch": ref_bytes_per_launch,
                     "reference_bytes_over_implemented_bytes": ref_bytes_per_launch / bytes_per_launch,
                     "queries_per_launch": mine,
                     "rows_on_8bit_copy_per_query": rows_code, "rows_on_f32_per_query": rows_f32,
                     "adjacency_words_per_query": adj_words, "screen_row_bytes": srow if rows_code > 0 else 0,
                     "counting_kernel_results_equal_timed_kernel": counters_equal,
                     "avg_cmps": ref_cmps, "avg_hops": ref_hops, "avg_scanned": ref_scan,
                     "kernel_cmps": lds_cmps, "grid": idx.info("last_grid"),
                     "kernel_ms_with_exact_on_chip_visited_set": exact_form_ms, "kernel_cmps_with_exact_on_chip_visited_set": exact_form_cmps,
                     "exact_on_chip_visited_set_cmps_equal_reference_for_every_query_of_batch_0": exact_form_equal,
                     "lds_bytes": idx.info("last_lds_bytes"), "ids_equal_exact_mode": same_ids,
                     # north_star asks for MFMA utilisation beside HBM GB/s: this kernel is GEMV shaped
                     # (0.5 flop/B) and issues no MFMA; the repo's MFMA kernel is the exact-kNN generator
                     "mfma_util": 0.0},
    }

    # HBM traffic per launch.  PMC counters cannot be read from inside the process, so rank 0 of a one-GPU run profiles
    # children of itself (`rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 bench.py --pmc-child ...`, the same for
    # WRITE_SIZE -- the two do not fit one pass -- and an L2 set), same data (seed), same graph (file), same L_pq, same rotating
    # batches: once with the launches of the regime `value` is measured in (depth streams, capped grid), once one launch at
    # a time; and one --kernel-trace pass (no counters: launches really overlap) for the span of the overlapped launches.
    if world == 1 and rank == 0 and not args.no_traffic:
        try:
            t0 = time.time()
            tr = measure_traffic(args, idx, chosen, mine, depth, pipe_grid)
            rf = result["roofline"]
            if tr.get("in_flight"):
                t_ = tr["in_flight"]
                rf.update({k_: v_ for k_, v_ in t_.items() if k_ != "traffic_source"})
                rf["traffic_source"] = t_["traffic_source"]
                rf["requested_bytes_frac_of_peak"] = t_["traffic"] / (eff_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                rf["frac_over_requested_bytes_frac"] = rf["frac"] / rf["requested_bytes_frac_of_peak"]
            if tr.get("serial"):
                t_ = tr["serial"]
                one = rf["one_launch_at_a_time"]
                one["traffic"] = t_["traffic"]
                one["l2_hit_share"] = t_.get("l2_hit_share")
                one["requested_bytes_frac_of_peak"] = t_["traffic"] / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                if rf["traffic"] is None:  # (no in-flight passes: the serial figure stands for both)
                    rf["traffic"], rf["l2_hit_share"] = t_["traffic"], t_.get("l2_hit_share")
                    rf["requested_bytes_frac_of_peak"] = t_["traffic"] / (eff_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                    rf["traffic_source"] = t_["traffic_source"]
            if tr.get("trace"):
                rf["rocprof_kernel_trace"] = tr["trace"]
            # Lead with what the memory side was asked for (PMC): `frac` priced on the implemented algorithm's bytes counts L2 hits as
            # HBM traffic (8-11 % of the line requests here); it stays beside it as `implemented_bytes_frac`.
            rf["implemented_bytes_achieved"], rf["implemented_bytes_frac"] = rf["achieved"], rf["frac"]
            one = rf["one_launch_at_a_time"]
            one["implemented_bytes_achieved"], one["implemented_bytes_frac"] = one["achieved"], one["frac"]
            if rf["traffic"]:
                rf["achieved"] = rf["traffic"] / (eff_ms * 1e-3) / 1e9
                rf["frac"] = rf["achieved"] / HBM_PEAK_GBS
                rf["frac_le_1"] = bool(rf["frac"] <= 1.0)
                rf["achieved_is"] = ("PMC traffic per 10 000-query launch (FETCH_SIZE x 2 + WRITE_SIZE, rocprofv3 child passes of this run over the per-launch "
                                     "kernel: a batch costs the same requests whichever way its queries are handed out) / kernel_ms")
            if one.get("traffic"):
                one["achieved"] = one["traffic"] / (kernel_ms * 1e-3) / 1e9
                one["frac"] = one["achieved"] / HBM_PEAK_GBS
            if rf["traffic"]:
                log(f"[rank 0] PMC traffic per launch {rf['traffic'] / 1e9:.2f} GB = {rf['traffic'] / bytes_per_launch:.3f} x implemented-algorithm "
                    f"bytes = {rf['traffic'] / ref_bytes_per_launch:.3f} x reference-priced bytes ({time.time() - t0:.0f}s for the profiled child passes)")
        except Exception as e:  # noqa: BLE001
            log(f"[rank 0] PMC traffic not measured in this run: {e}")
    try:
        import glob

        # No PMC passes in this run (N > 1: a rank cannot profile children of itself beside the others; --no-traffic): the newest committed
        # run of the same workload, L_pq and graph (its hash) gives the PMC traffic PER QUERY -- a query costs the same requests whichever
        # way its batch is cut -- times this rank's queries per step; the line says where the figure is from.
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_search_kernel_summary.json")))[::-1]:
            rf = result["roofline"]
            if rf["traffic"] is not None:
                break
            bl = json.load(open(f))["bench_line"]
            c, rp = bl["config"], bl["roofline"]
            if (c["workload"] == result["config"]["workload"] and c["L_pq"] == chosen and c.get("graph_hash") == result["config"]["graph_hash"]
                    and rp.get("traffic") and rp.get("queries_per_launch")):
                rf["traffic"] = rp["traffic"] / rp["queries_per_launch"] * mine
                rf["traffic_source"] = (f"{os.path.relpath(f, ROOT)}: the PMC traffic per query of that run (same workload, L_pq, graph hash; rocprofv3 FETCH_SIZE x 2 + "
                                        f"WRITE_SIZE) x this rank's {mine} queries per step -- no PMC passes in this run")
                rf["implemented_bytes_achieved"], rf["implemented_bytes_frac"] = rf["achieved"], rf["frac"]
                rf["achieved"] = rf["traffic"] / (eff_ms * 1e-3) / 1e9
                rf["frac"] = rf["requested_bytes_frac_of_peak"] = rf["achieved"] / HBM_PEAK_GBS
                rf["frac_le_1"] = bool(rf["frac"] <= 1.0)
                rf["achieved_is"] = "PMC traffic per query of a committed run of the same workload x this rank's queries per step / kernel_ms"
                break
    except Exception as e:  # noqa: BLE001
        log(f"[rank {rank}] no PMC traffic figure: {e}")
    if not result["roofline"]["frac_le_1"]:
        log(f"[rank {rank}] ERROR: roofline.frac = {result['roofline']['frac']:.3f} > 1: the launch cannot have moved {result['roofline']['achieved'] * eff_ms * 1e-3:.2f} GB "
            f"from HBM in {eff_ms:.3f} ms -- cache hits (small set?) or a counting error")

    if world == 1 and rank == 0 and not args.no_host_path:
        result.update(host_buffer_path(np, torch, idx, batches, outs, args, chosen))
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(np, torch, idx, base, batches[0], outs[0], args, chosen, metric)
    if world == 1 and rank == 0 and not args.no_other_workloads and (args.workload, args.n) == ("t2i", 10_000_000) and not args.sweep:
        result["other_workloads"] = run_other_workloads(args)

    if rank == 0:
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.destroy_process_group()


def compact_workload(line):
    """What `other_workloads` keeps of a full bench line of another shape (BASELINE configs[0] / configs[2] run by the same code)."""
    c, r = line["config"], line["roofline"]
    one = r.get("one_launch_at_a_time", {})
    cb = line.get("cpu_baseline") or {}
    return {"workload": c["workload"], "metric": line["metric"], "value": line["value"], "value_regime": c["value_regime"],
            "value_one_launch_at_a_time": c["value_one_launch_at_a_time"], "value_batch_stream": c.get("value_batch_stream"),
            "value_launches_in_flight": c.get("value_launches_in_flight"), "ms_per_step": line["ms_per_step"], "steps": line["steps"],
            "L_pq": c["L_pq"], "recall_at_10": c["recall_at_10"], "n_base": c["n_base"], "dim": c["dim"], "metric_kind": c["metric"],
            "screen": c["screen"] != "off", "screen_pass_rate": c.get("screen_pass_rate"), "visited": c["visited"],
            "roofline_frac": r["frac"], "roofline_frac_one_launch_at_a_time": one.get("frac"), "kernel_ms_one_launch_at_a_time": one.get("kernel_ms"),
            "traffic_bytes_per_launch": r.get("traffic"), "implemented_bytes_per_launch": r["bytes_per_launch"],
            "traffic_over_implemented_bytes": (r["traffic"] / r["bytes_per_launch"]) if r.get("traffic") else None,
            "requested_bytes_frac_of_peak_one_launch_at_a_time": one.get("requested_bytes_frac_of_peak"),
            "kernel_cmps_over_avg_cmps": (r["kernel_cmps"] / r["avg_cmps"]) if r.get("avg_cmps") else None,
            "cpu_baseline_value": cb.get("value"), "cpu_baseline_cores": cb.get("cores"),
            "gpu_vs_cpu_ids_identical_fraction": cb.get("gpu_vs_cpu_ids_identical_fraction"),
            "index": c["index"], "build_seconds": c["build_seconds"]}


def run_other_workloads(args):
    """BASELINE configs[0] (SIFT-1M shape, 128-d L2) and configs[2] (CLIP-WebVid shape, 2.5 M x 512-d IP; shapes:
    prepare_for_clip_webvid.py:89-128, run_roargraph_search_test.sh:8-15) through this same script as child processes -- fewer steps
    and batches, a shorter CPU sample, no host-buffer timing, no share-of-8 -- each reduced to a compact object."""
    import subprocess

    out = {}
    for w in ("webvid", "sift"):
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", w, "--steps", "10", "--warmup", "2", "--batches", "2",
               "--cpu-seconds", "5", "--no-host-path", "--no-share-probe", "--no-other-workloads", "--seed", str(args.seed)]
        t0 = time.time()
        try:
            pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            lines = [x for x in pr.stdout.decode(errors="replace").splitlines() if x.startswith("{")]
            if pr.returncode != 0 or not lines:
                raise RuntimeError(f"rc {pr.returncode}: {pr.stderr.decode(errors='replace')[-300:]}")
            out[w] = compact_workload(json.loads(lines[-1]))
            out[w]["seconds"] = time.time() - t0
            log(f"[rank 0] other workload {w}: {out[w]['value'] / 1e6:.2f} M QPS at L_pq {out[w]['L_pq']} (recall {out[w]['recall_at_10']:.4f}) in {time.time() - t0:.0f}s")
        except Exception as e:  # noqa: BLE001
            out[w] = {"error": str(e)[-400:]}
            log(f"[rank 0] other workload {w} failed: {e}")
    return out


def counter_per_launch(csv_path, counter, steps):
    """Mean of `counter` over the last `steps` dispatches of the search kernel in a rocprofv3 counter_collection.csv (the
    timed loop is the last thing the profiled child launches; the builder's searches run under their own kernel name)."""
    import csv

    rows = [x for x in csv.DictReader(open(csv_path))
            if "roar_search_v2_kernel" in x["Kernel_Name"] and "build" not in x["Kernel_Name"]
            and x.get("Counter_Name", counter) == counter]
    if len(rows) < steps:
        raise RuntimeError(f"{len(rows)} profiled launches of the search kernel in {csv_path}, expected at least {steps}")
    return sum(float(x["Counter_Value"]) for x in rows[-steps:]) / steps


def measure_traffic(args, idx, L, nq_launch, depth=1, pipe_grid=0):
    """FETCH_SIZE / WRITE_SIZE (+ an L2 set) of the search kernel's timed launches, from profiled child runs of this script
    (MI355X_MICROARCH.md: separate --pmc passes with --kernel-trace only; FETCH_SIZE counts 128-B requests at 64 B on
    gfx950 -> x2, calibrated on this access pattern in profiles/r01_fetch_calibration.txt).  Two regimes: the one `value` is
    measured in (`depth` streams, at most `pipe_grid` workgroups per launch; a --pmc pass serialises dispatches, so what it
    sees is the capped launch alone) and one launch at a time; plus one --kernel-trace pass without counters, where the
    launches really overlap: their span.  Children are started as child processes (never exec'ed over this one), the interpreter
    itself after `--`.  -> {"in_flight": {...} | None, "serial": {...}, "trace": {...} | None}"""
    import glob, shutil, subprocess, tempfile

    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        raise RuntimeError("rocprofv3 not found")
    own_cache = not (args.graph_cache and os.path.exists(args.graph_cache))
    cache = args.graph_cache
    if own_cache:
        fd, cache = tempfile.mkstemp(prefix="roar_bench_graph_", suffix=".index", dir="/tmp")
        os.close(fd)
        idx.SaveProjectionGraph(cache)
    steps = 8
    # the launch plan of the timed loop (what the calibration picked in THIS process): screen on / off, visited form, tags
    vm = idx.info("last_visited_mode")
    plan = f"{idx.info('last_screen')},{0 if vm == 0 else (3 if idx.info('last_visited_forgetful') else 1)},{idx.info('last_vis_entries') if vm != 0 else 0}"
    child0 = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--graph-cache", cache, "--L", str(L), "--workload", args.workload,
              "--n", str(args.n), "--dim", str(args.dim), "--nq", str(args.nq), "--batches", str(args.batches), "--k", str(args.k),
              "--seed", str(args.seed), "--visited", args.visited, "--vis-entries", str(args.vis_entries), "--steps", str(steps),
              "--warmup", "1", "--builder", args.builder, "--learn", "0", "--ref-learn", "1", "--ref-learn2", "0",
              "--hard-offset", str(args.hard_offset), "--plan", plan] + (["--no-screen"] if args.no_screen else [])
    env = dict(os.environ, TMPDIR="/tmp")
    L2SET = ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_DRAM_sum")

    def profiled(child, names):
        """one rocprofv3 run of `child`; names = counters of a --pmc pass, or () for --kernel-trace alone -> output dir"""
        out = tempfile.mkdtemp(prefix="roar_pmc_", dir="/tmp")
        cmd = [prof, "--kernel-trace"] + (["--pmc"] + list(names) if names else []) + ["--output-format", "csv", "-d", out, "--"] + child
        # own session: a timeout must end the profiler AND the profiled child, not just the former
        pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
        try:
            _, err = pr.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            import signal

            os.killpg(pr.pid, signal.SIGKILL)  # (the exact process group started above)
            pr.communicate()
            shutil.rmtree(out, ignore_errors=True)
            raise RuntimeError(f"rocprofv3 {names or 'kernel-trace'} timed out")
        if pr.returncode != 0:
            shutil.rmtree(out, ignore_errors=True)
            raise RuntimeError(f"rocprofv3 {names or 'kernel-trace'} failed (rc {pr.returncode}): {err.decode(errors='replace')[-300:]}")
        return out

    def counters(child, what):
        vals = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE", L2SET):
            names = [counter] if isinstance(counter, str) else list(counter)
            try:
                out = profiled(child, names)
            except RuntimeError as e:
                if counter is L2SET:  # optional: a box whose profiler refuses the set just leaves these fields out
                    log(f"[rank 0] L2 counter pass not available here: {e}")
                    continue
                raise
            try:
                files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
                if not files:
                    if counter is L2SET:
                        continue
                    raise RuntimeError(f"rocprofv3 --pmc {counter}: no counter file")
                for nm in names:
                    try:
                        vals[nm] = counter_per_launch(max(files, key=os.path.getmtime), nm, steps)  # (SIZE counters: KB per launch)
                    except RuntimeError:
                        if counter is not L2SET:
                            raise
            finally:
                shutil.rmtree(out, ignore_errors=True)
        extra = {}
        if "TCC_HIT_sum" in vals and "TCC_MISS_sum" in vals and vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"] > 0:
            extra["l2_hit_share"] = vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"])  # of 128-B line requests at the L2
            extra["beyond_l2_bytes"] = vals["TCC_MISS_sum"] * 128.0  # what the L2 asks the fabric for: Infinity Cache + HBM together
            extra["l2_requests_per_launch"] = vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]
        if "TCC_EA0_RDREQ_sum" in vals and vals.get("TCC_EA0_RDREQ_sum", 0) > 0 and "TCC_EA0_RDREQ_DRAM_sum" in vals:
            extra["l2_read_requests_routed_to_memory_controllers_share"] = vals["TCC_EA0_RDREQ_DRAM_sum"] / vals["TCC_EA0_RDREQ_sum"]
        if extra:
            # the Infinity Cache (256 MB, memory side) sits BEHIND these counters: a hit there is still an L2 miss and a
            # memory-side read request, so HBM-only bytes are not separable with the counters rocprofv3 lists on gfx950
            extra["frac_hbm_only"] = None
        return {"traffic": vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024, **extra,
                "traffic_source": f"measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE over {steps} launches of a "
                                  f"child of this command (same data, graph, L_pq, batches; {what}); FETCH_SIZE x 2 (gfx950)",
                "FETCH_SIZE_KB_raw_per_launch": vals["FETCH_SIZE"], "WRITE_SIZE_KB_raw_per_launch": vals["WRITE_SIZE"]}

    res = {"in_flight": None, "serial": None, "trace": None}
    try:
        flight = ["--in-flight", str(depth), "--launch-grid", str(pipe_grid)]
        if depth > 1:
            res["in_flight"] = counters(child0 + flight, f"launch plan of the timed loop: at most {pipe_grid} workgroups; the counter "
                                                          f"passes serialise the launches")
        res["serial"] = counters(child0 + ["--in-flight", "1"], "one launch at a time, uncapped grid")
        try:  # launches overlapping for real: start / end of every launch from the kernel trace
            out = profiled(child0 + (flight if depth > 1 else ["--in-flight", "1"]), ())
            try:
                files = glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True)
                span, mean = kernel_trace_span(max(files, key=os.path.getmtime), steps)
                res["trace"] = {"launches": steps, "in_flight": depth, "span_ms": span, "span_ms_per_launch": span / steps,
                                "mean_launch_duration_ms": mean}
            finally:
                shutil.rmtree(out, ignore_errors=True)
        except Exception as e:  # noqa: BLE001
            log(f"[rank 0] kernel-trace pass of the overlapped loop not available: {e}")
    finally:
        if own_cache and os.path.exists(cache):
            os.remove(cache)
    return res


def host_buffer_path(np, torch, idx, batches, outs, args, L):
    """The drop-in call itself -- roar_search_batch with HOST buffers, what tests/test_search_roargraph.cpp:205 becomes:
    queries in pinned memory (as the facade's load_data / data_align return them) read in place by the kernel, results
    copied back; and the same with an ordinary pageable query block (packed into the library's pinned block first).
    Rotating batches, wall clock over `steps` calls (each call returns with its results on the host)."""
    import ctypes as C

    from mysteryann_amd import capi

    lib = capi.lib()
    B = len(batches)
    nq, d = batches[0].shape
    pinned, pageable = [], []
    for b in range(B):
        hq = batches[b].cpu().numpy()
        ptr = lib.roar_host_alloc(hq.nbytes)
        blk = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=hq.shape)
        blk[:] = hq
        pinned.append((ptr, blk))
        pageable.append(np.ascontiguousarray(hq))
    ids = np.zeros((nq, args.k), np.uint32)
    dists = np.zeros((nq, args.k), np.float32)
    cmps = np.zeros(nq, np.uint32)
    hops = np.zeros(nq, np.uint32)

    def call(q):
        capi.check(lib.roar_search_batch(idx._h, capi.f32p(q), nq, d, args.k, L, capi.u32p(ids), capi.f32p(dists),
                                         capi.u32p(cmps), capi.u32p(hops), None))

    res = {}
    for name, blocks in (("value_host_buffers", [p[1] for p in pinned]), ("value_host_buffers_pageable_queries", pageable)):
        for s_ in range(args.warmup):
            call(blocks[s_ % B])
        t0 = time.perf_counter()
        for s_ in range(args.steps):
            call(blocks[s_ % B])
        res[name] = nq * args.steps / (time.perf_counter() - t0)
    last = (args.steps - 1) % B
    res["host_buffers_ids_equal_device_path"] = bool((ids.view(np.int32) == outs[last]["ids"].cpu().numpy()).all())
    for ptr, _ in pinned:
        lib.roar_host_free(ptr)
    return res


def run_sharded(args, np, torch, mi, synth, dev, rank, world, use_dist):
    """Index-sharded configuration (BASELINE configs[4]): rank r holds base rows [r * m, (r + 1) * m) of one
    T2I-shape distribution and its own graph over them (local ids); a step = every rank searches the whole query block on
    its shard, ids are rebased to global, per-shard top-k lists are all-gathered (RCCL over xGMI, nq x k x 8 B per rank)
    and merged by (distance, id) on the GPU.  value = queries per second of that whole step, max over ranks."""
    import torch.distributed as dist

    from mysteryann_amd import parallel

    d, k, nq = 200, args.k, args.nq
    m = args.n_explicit or 12_500_000  # points per shard
    n_learn = args.learn if args.learn >= 0 else (m // 5 if args.builder == "reference" else m // 10)
    t0 = time.time()
    A, _, sc = synth._maps(d, args.seed)  # one distribution for all shards, own latent draws per shard
    g = torch.Generator(device=dev)
    g.manual_seed(5000 + rank)
    base = torch.empty((m, d), device=dev)
    At = torch.from_numpy(A).to(dev)
    for i in range(0, m, 1 << 20):
        mm = min(1 << 20, m - i)
        base[i:i + mm] = torch.randn((mm, synth.LATENT), generator=g, device=dev) @ At \
            + (0.3 * float(sc)) * torch.randn((mm, d), generator=g, device=dev)
    _, queries, learn = synth.t2i_torch(1024, nq, d, seed=args.seed, n_learn=n_learn, device=dev)  # same queries on every rank
    idx = mi.IndexBipartite(d, m, mi.INNER_PRODUCT, None, device=dev.index or 0)
    idx.SetBaseData(base)
    if args.builder == "reference":  # exact lists of the training queries on this shard -> LinkProjection, canonical parameters
        lists, _ = idx.ExactKnnDevice(learn, 100, want_dists=False)
        prm = mi.Parameters()
        prm.Set("M_sq", 100); prm.Set("M_pjbp", 35); prm.Set("L_pjpq", 500)
        idx.BuildRoarGraphDevice(lists, prm)
        del lists
    else:
        idx.BuildBootstrapGraph(learn=learn, max_degree=args.build_maxdeg, row_keep=args.build_keep)
    torch.cuda.synchronize()
    del learn
    log(f"[rank {rank}] shard of {m} points generated and indexed in {time.time() - t0:.1f}s, avg degree "
        f"{idx.info('nedges') / m:.1f}")

    def gather(ids, dd):
        if use_dist:
            return parallel.allgather_topk(ids, dd)
        return ids[None].contiguous(), dd[None].contiguous()

    def step(L):
        out = idx.SearchRoarGraphBatchDevice(queries, k, L)
        ia, da = gather(parallel.rebase_ids(out["ids"], rank * m), out["dists"])
        return parallel.merge_topk_device(ia, da), out

    gi, gd = idx.ExactKnnDevice(queries, k)  # exact global ground truth: merge of exact per-shard lists
    ga, gda = gather(parallel.rebase_ids(gi, rank * m), gd)
    gt = parallel.merge_topk_device(ga, gda)[0].to(torch.int64)
    chosen = args.L
    if chosen == 0:
        for L in L_LIST:
            if L < k or L > args.max_L:
                continue
            (mi_, _), _ = step(L)
            torch.cuda.synchronize()
            r = recall_at_k(torch, mi_, gt, k)
            log(f"[rank {rank}] L_pq={L:5d} global recall@{k}={r:.4f} shard kernel={idx.info('last_kernel_ns') / 1e6:.3f} ms")
            if r >= args.target_recall:
                chosen = L
                break
        chosen = chosen or L_LIST[-1]
    (mi_, md_), out = step(chosen)
    torch.cuda.synchronize()
    rec = recall_at_k(torch, mi_, gt, k)
    idx.set_option("visited_mode", 0)  # reference-algorithm counters for the algorithmic bytes, untimed
    ex = idx.SearchRoarGraphBatchDevice(queries, k, chosen)
    torch.cuda.synchronize()
    idx.set_option("visited_mode", 2)
    dpad = idx.info("dim_padded")
    ref_bytes = (4 * dpad * ex["cmps"].double().sum() + 4 * ex["scanned"].double().sum() + 8 * ex["hops"].double().sum()
                 + (4 * dpad + 8 * k) * nq).item()
    tblock = torch.zeros((nq, 4), dtype=torch.int32, device=dev)  # what the timed launch itself fetches (counting instantiation)
    idx.SetTrafficCounters(tblock)
    idx.SearchRoarGraphBatchDevice(queries, k, chosen)
    torch.cuda.synchronize()
    idx.SetTrafficCounters(None)
    tb = tblock.to(torch.float64).sum(dim=0)
    alg_bytes = implemented_bytes(tb[0].item(), tb[1].item(), tb[2].item(), nq, dpad, idx.info("screen_row_bytes"), k)
    for _ in range(args.warmup):
        step(chosen)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(chosen)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kernel_ms = idx.info("last_kernel_ns") * 1e-6
    if use_dist:
        t = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = t.item()
    ms_per_step = wall * 1e3 / args.steps
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    result = {
        "metric": "QPS @ recall@10=0.9, Text-to-Image-100M index-sharded" if world * m == 100_000_000
        else f"QPS @ recall@10={args.target_recall}, t2i shape index-sharded",
        "value": nq * args.steps / wall, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"t2i-{world * m // 1_000_000}M-ip-index-sharded", "n_base": world * m, "per_shard": m, "dim": d,
                   "metric": "ip", "n_queries": nq, "k": k, "L_pq": chosen, "recall_at_10": rec,
                   "parallelism": f"base + graph sharded {world}-way, all queries on every shard, RCCL all-gather of "
                                  f"{nq * k * 8} B per rank + HIP top-k merge",
                   "builder": args.builder, "avg_degree": idx.info("nedges") / m},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "kernel": "roar_search_v2_kernel (this rank's shard)", "kernel_ms": kernel_ms,
                     "regime": "one launch per step on this rank's shard: kernel_ms = HIP events around the last launch",
                     "frac_le_1": bool(achieved / HBM_PEAK_GBS <= 1.0),
                     "exchange_and_merge_ms": ms_per_step - kernel_ms, "bytes_per_launch": alg_bytes,
                     "bytes": "IMPLEMENTED algorithm (kernel's own counters: rows on the 8-bit copy, rows on f32, adjacency words)",
                     "reference_bytes_speedup": ref_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "reference_bytes_per_launch": ref_bytes,
                     "queries_per_launch": nq, "mfma_util": 0.0},
    }
    if rank == 0:
        print(json.dumps(result), flush=True)


def effective_cpus():
    """Host cores this process may really use: min(affinity, cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(np, torch, idx, base, myq, out, args, L, metric):
    """The reference's loop shape (100 warm-up queries, then omp parallel for schedule(dynamic,1),
    tests/test_search_roargraph.cpp:198-212) on the host cores, same graph, same queries."""
    import oracle

    impl = "ref" if oracle.have_ref() else "port"
    t0 = time.time()
    hb = base.cpu().numpy()
    ep, off, nbrs = idx.GetProjectionGraph()
    hq = myq.cpu().numpy()
    if hq.shape[1] != hb.shape[1]:
        hq = np.ascontiguousarray(np.pad(hq, ((0, 0), (0, hb.shape[1] - hq.shape[1]))))
    log(f"[cpu] host copies of base + graph in {time.time() - t0:.1f}s; impl={impl}")
    # the GPU box runs this container under a cgroup CPU quota (16 of the host's 256 hardware threads at
    # the time of writing): more threads than the quota only thrash, so use exactly the quota.  16 is
    # also the reference's canonical -T (run_roargraph_search_test.sh:1).
    threads = min(oracle.max_threads(impl), effective_cpus())
    dpad = idx.info("dim_padded")
    oracle.search_batch(hb, off, nbrs, ep, hq[:100], args.k, L, metric, threads=threads, impl=impl, dim=dpad)  # warm-up
    done, spent, reps, res = 0, 0.0, 0, None
    while spent < args.cpu_seconds and reps < 50:
        t0 = time.perf_counter()
        res = oracle.search_batch(hb, off, nbrs, ep, hq, args.k, L, metric, threads=threads, impl=impl, dim=dpad)
        spent += time.perf_counter() - t0
        done += hq.shape[0]
        reps += 1
    gpu_ids = out["ids"].cpu().numpy().view(np.uint32)
    gpu_d = out["dists"].cpu().numpy()
    ids_equal = float((gpu_ids == res["ids"]).all(axis=1).mean())
    return {"value": done / spent, "unit": "queries/s", "cores": threads, "host_threads_available": os.cpu_count(),
            "host_threads_usable_under_cgroup_quota": effective_cpus(),
            "kind": "reference" if impl == "ref" else "port",
            "sample": f"all {hq.shape[0]} queries x {reps} passes at L_pq={L}, omp dynamic,1 over {threads} threads "
                      f"(= cgroup CPU quota; host has {os.cpu_count()} hardware threads) "
                      f"({'reference distance/queue/visited headers compiled in place, search loop restated' if impl == 'ref' else 'C restatement'})",
            "gpu_vs_cpu_ids_identical_fraction": ids_equal,
            "gpu_vs_cpu_max_abs_dist_diff": float(np.abs(gpu_d - res["dists"]).max())}


if __name__ == "__main__":
    main()
