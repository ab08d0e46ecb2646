"""Round 5: where does a stream of SMALL batches lose against a stream of large ones?  Same index, same L_pq, the batch-stream form
(roar_search_stream_*): batch size x total queries x which rows, steady-state hand-out rate from the device clock (first query of
batch 3 taken .. first query of the last batch taken), wall per step, and the same through per-launch calls in flight."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from _graph import t2i_index

L, K, B = int(os.environ.get("L_PQ", "50")), 10, 8
idx, base, queries = t2i_index(10_000_000, 10_000 * B, "reference", 0, "/tmp/roar_graph.index")
dev = queries.device


def bufs(rows, n):
    return [dict(ids=torch.empty((rows, K), dtype=torch.int32, device=dev), dists=torch.empty((rows, K), dtype=torch.float32, device=dev),
                 cmps=torch.empty(rows, dtype=torch.int32, device=dev), hops=torch.empty(rows, dtype=torch.int32, device=dev),
                 scanned=torch.empty(rows, dtype=torch.int32, device=dev)) for _ in range(n)]


def rows_of(step, bs, mode):
    if mode == "quarters":
        per = 10_000 // bs
        lo = ((step // per) % B) * 10_000 + (step % per) * bs
    elif mode == "share":  # bench.py's emulated share: the first bs rows of batch (step % B)
        lo = (step % B) * 10_000
    else:  # "all": walk through all 80 000 distinct queries
        lo = (step * bs) % (10_000 * B - bs + 1)
    return queries[lo:lo + bs]


_keep = {}


def run(bs, total, mode):
    steps = total // bs
    if (bs, steps) not in _keep:
        if mode == "quarters":  # the batches are consecutive pieces of 10 000-row blocks: byte for byte the addresses of the bs = 10 000 run
            big = bufs(10_000, total // 10_000)
            per = 10_000 // bs
            _keep[(bs, steps)] = [{k_: v_[(s_ % per) * bs:(s_ % per + 1) * bs] for k_, v_ in big[s_ // per].items()} for s_ in range(steps)]
        else:
            _keep[(bs, steps)] = bufs(bs, steps)
    ob = _keep[(bs, steps)]
    st = idx.OpenSearchStream(K, L)
    try:
        tk = [st.submit(rows_of(s, bs, mode), ob[s])[0] for s in range(min(3, steps))]
        [st.wait(t) for t in tk]
        t_end = time.time() + 2
        while st.info("workgroups_alive") and time.time() < t_end:
            time.sleep(0.001)
        torch.cuda.synchronize()
        l0 = st.info("launches")
        tn0, tc0, tb0 = st.info("take_ns_total"), st.info("take_calls"), st.info("busy_ns_total")
        t0 = time.perf_counter()
        lead = None
        if os.environ.get("LEAD"):  # one 10 000-query batch in front: the whole stream is then served by ONE launch of workgroups
            lead = st.submit(queries[:10_000])[0]
        tk = [st.submit(rows_of(s, bs, mode), ob[s])[0] for s in range(steps)]
        t_sub = time.perf_counter() - t0
        alive_mid = st.info("workgroups_alive")
        if lead:
            st.wait(lead)
        if os.environ.get("HOSTSLEEP"):  # the host stays away from the completion records while the GPU works
            time.sleep(0.06)
        [st.wait(t) for t in tk]
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        tt = [st.times(t) for t in tk]
        tn1, tc1, tb1 = st.info("take_ns_total"), st.info("take_calls"), st.info("busy_ns_total")
        span = (max(b for _, b in tt) - min(a for a, _ in tt)) * 1e-3
        k0 = min(3, steps - 2)
        steady = (tt[-1][0] - tt[k0][0]) * 1e-3 / max(1, steps - 1 - k0) if steps > k0 + 1 else float("nan")
        drain = (tt[-1][1] - tt[-1][0]) * 1e-3
        print(f"bs={bs:6d} total={total:7d} rows={mode:5s}: wall {wall * 1e6:8.0f} us = {wall * 1e3 / steps:.4f} ms/step = {total / wall / 1e6:.2f} M QPS | device span {span:8.0f} us | "
              f"steady hand-out {steady:7.1f} us/batch = {bs / steady:.2f} M QPS | last batch taken..done {drain:.0f} us | submits {t_sub * 1e6:.0f} us, "
              f"launches {st.info('launches') - l0}, alive after submits {alive_mid}, heals {st.info('heals')}, "
              f"dequeue {(tn1 - tn0) / max(1, tc1 - tc0) / 1e3:.1f} us per call over {tc1 - tc0} calls, search {(tb1 - tb0) / total / 1e3:.1f} us per query, "
              f"workgroup-time accounted {((tn1 - tn0) + (tb1 - tb0)) / 1e3 / 3072:.0f} us of span", flush=True)
    finally:
        st.close()


if True:
    for mg in [int(x) for x in os.environ.get("MAXGRID", "0").split(",")]:
        idx.set_option("max_grid", mg)
        print(f"# L_pq={L} max_grid={mg}", flush=True)
        for bs in [int(x) for x in os.environ.get('BS', '10000,2500,1250').split(',')] * int(os.environ.get('REPEAT', '1')):
            run(bs, 200_000, os.environ.get("MODE", "share"))
