"""How do the host cores of the GPU box behave for the CPU baseline?  (thread scaling probe)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, oracle
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, e)
rng = np.random.default_rng(0)
n, d, nq = 2_000_000, 200, 4000
base = rng.standard_normal((n, d), dtype=np.float32)
q = rng.standard_normal((nq, d), dtype=np.float32)
deg = 32
off = (np.arange(n + 1, dtype=np.uint64) * deg)
nbrs = rng.integers(0, n, n * deg, dtype=np.uint32)
for impl in (["ref"] if oracle.have_ref() else []) + ["port"]:
    for th in (1, 16, 64, 128, 256):
        k = max(1, nq * th // 256)
        oracle.search_batch(base, off, nbrs, 0, q[:64], 10, 75, 1, threads=th, impl=impl)
        t0 = time.perf_counter(); r = oracle.search_batch(base, off, nbrs, 0, q[:k], 10, 75, 1, threads=th, impl=impl); dt = time.perf_counter() - t0
        print(impl, "threads", th, "queries", k, f"qps {k/dt:.0f} cmps/q {r['cmps'].mean():.0f} per-thread cmps/s {r['cmps'].sum()/dt/th:.0f}", flush=True)
