"""What one of N GPUs sees in the strong-scaling run (BASELINE configs[3]): a 10 000 / N-query slice of every batch, L_pq = 55,
default graph.  Per slice size: kernel time of one launch (HIP events of the library), wall per step of a loop of launches
one after the other, and of the same loop with 2 / 3 launches in flight on alternating streams (bench.py --in-flight)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # as bench.py: one hardware queue per stream (read when the HIP runtime starts)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from _graph import t2i_index

L, K, B, STEPS = int(os.environ.get("L_PQ", "55")), 10, 8, 64
idx, base, queries = t2i_index(10_000_000, 10_000 * B, "reference", 0, "/tmp/roar_graph.index")
dev = queries.device
main = torch.cuda.current_stream(dev)
side = [torch.cuda.Stream(dev) for _ in range(16)]


def loop(batches, outs, depth):
    streams = [main] if depth == 1 else side[:depth]
    for s in range(4):
        idx.SearchRoarGraphBatchDevice(batches[s % B], K, L, out=outs[s % B], stream=streams[s % depth])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(STEPS):
        idx.SearchRoarGraphBatchDevice(batches[s % B], K, L, out=outs[s % B], stream=streams[s % depth])
    host = (time.perf_counter() - t0) * 1e3 / STEPS  # the host's share: all launches enqueued, nothing waited for yet
    torch.cuda.synchronize()
    loop.host_ms = min(getattr(loop, "host_ms", 1e9), host)
    return (time.perf_counter() - t0) * 1e3 / STEPS


ref = None
GS = [int(x) for x in os.environ.get("ONLY_N", "1,2,4,8,16").split(",")]
DEPTHS = [int(x) for x in os.environ.get("DEPTHS", "1,2,4,6,8").split(",")]
for g in GS:
    nq = 10_000 // g
    batches = [queries[b * 10_000:b * 10_000 + nq].contiguous() for b in range(B)]
    outs = [idx.SearchRoarGraphBatchDevice(batches[b], K, L) for b in range(B)]
    torch.cuda.synchronize()
    best = 1e9
    for r in range(6):
        idx.SearchRoarGraphBatchDevice(batches[r % B], K, L, out=outs[r % B])
        torch.cuda.synchronize()
        best = min(best, idx.info("last_kernel_ns") / 1e6)
    hops = torch.stack([o["hops"] for o in outs]).float()
    row = [f"N={g:2d} nq/GPU={nq:5d} grid {idx.info('last_grid'):4d} kernel {best:.4f} ms  hops mean {hops.mean():.1f} max {hops.max():.0f}"]
    MG = [int(x) for x in os.environ.get("MAXGRID", "0").split(",")]
    for depth, mg in [(x, y) for x in DEPTHS for y in MG]:  # negative: without the library's timing events
        idx.set_option("launch_events", 1 if depth > 0 else 0)
        depth = abs(depth)
        idx.set_option("max_grid", mg)
        ms = min(loop(batches, outs, depth) for _ in range(3))
        if ref is None:
            ref = ms
        row.append(f"in-flight {depth}" + (f" grid<={mg}" if mg else "") + ("" if idx.info("launch_events_on") else " (no events)") + f": {ms:.4f} ms/step ({ref / ms:.2f}x)")
    idx.set_option("max_grid", 0)
    row.append(f"host enqueue {loop.host_ms * 1e3:.0f} us/launch")
    loop.host_ms = 1e9
    print("  ".join(row), flush=True)
