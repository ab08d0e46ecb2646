"""Round 6: which on-chip tag table serves which REGIME?  The calibration times candidates one launch at a time on the sample; the batch
stream (and launches in flight) live on resident workgroups.  Per table size (calibrate = 0, screen on, forgetful table): the one-launch
kernel time (HIP events, median and best of 7, so the calibration's noise is visible) and the stream's ms per step for 10 000-query
and 1 250-query batches (24 steps, best and median of 3 runs).  T2I-10M, L_pq from L_PQ (default 50).
  python scripts/table_regime_probe.py [tags ...]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from _graph import t2i_index

L, K, B, STEPS = int(os.environ.get("L_PQ", "50")), 10, 8, 24
idx, base, queries = t2i_index(10_000_000, 10_000 * B, "reference", 0, "/tmp/roar_graph.index")
dev = queries.device
tables = [int(x) for x in sys.argv[1:]] or [2048, 4096, 8192]


def bufs(rows, n):
    return [dict(ids=torch.empty((rows, K), dtype=torch.int32, device=dev), dists=torch.empty((rows, K), dtype=torch.float32, device=dev),
                 cmps=torch.empty(rows, dtype=torch.int32, device=dev), hops=torch.empty(rows, dtype=torch.int32, device=dev),
                 scanned=torch.empty(rows, dtype=torch.int32, device=dev)) for _ in range(n)]


def one_launch(reps=7):
    ts = []
    for r in range(reps + 1):
        idx.SearchRoarGraphBatchDevice(queries[(r % B) * 10_000:(r % B + 1) * 10_000], K, L)
        torch.cuda.synchronize()
        ts.append(idx.info("last_kernel_ns") / 1e6)
    ts = ts[1:]
    return float(np.median(ts)), min(ts)


def stream(bs, ob):
    st = idx.OpenSearchStream(K, L)
    try:
        tk = [st.submit(queries[(s % B) * 10_000:(s % B) * 10_000 + bs], ob[s])[0] for s in range(3)]
        [st.wait(t) for t in tk]
        t_end = time.time() + 2
        while st.info("workgroups_alive") and time.time() < t_end:
            time.sleep(0.0005)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tk = [st.submit(queries[(s % B) * 10_000:(s % B) * 10_000 + bs], ob[s])[0] for s in range(STEPS)]
        [st.wait(t) for t in tk]
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        info = (st.info("vis_entries"), st.info("resident_capacity"), st.info("lds_bytes"))
    finally:
        st.close()
    return wall * 1e3 / STEPS, info


big, small = bufs(10_000, STEPS), bufs(1_250, STEPS)
for rnd in range(int(os.environ.get("ROUNDS", "2"))):  # (interleaved rounds in one process)
    for tags in tables:
        idx.set_option("calibrate", 0), idx.set_option("screen", 2), idx.set_option("visited_mode", 3), idx.set_option("vis_entries", tags)
        med, best = one_launch()
        s10 = sorted(stream(10_000, big)[0] for _ in range(3))
        s1, info = zip(*[stream(1_250, small) for _ in range(3)])
        s1 = sorted(s1)
        print(f"round {rnd} L_pq {L} tags {tags:6d} (stream: {info[0][0]} tags, {info[0][1]} workgroups, {info[0][2]} B LDS): one launch median {med:.4f} best {best:.4f} ms | "
              f"stream 10 000 per batch best {s10[0]:.4f} median {s10[1]:.4f} ms/step | 1 250 per batch best {s1[0]:.4f} median {s1[1]:.4f} ms/step", flush=True)
