"""First search of a fresh index in EXACT visited mode with the calibration on: cmps must equal the oracle's (not `scanned`)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from mysteryann_amd import index as mi
rng = np.random.default_rng(3)
n, d, nq = 20000, 200, 1000
base = rng.standard_normal((n, d)).astype(np.float32)
q = rng.standard_normal((nq, d)).astype(np.float32)
deg = rng.integers(5, 60, n); off = np.zeros(n + 1, np.uint64); off[1:] = np.cumsum(deg)
nbrs = rng.integers(0, n, int(off[-1])).astype(np.uint32)
o = oracle.search_batch(base, off, nbrs, 3, q, 10, 20, 1, threads=8)
for cal in (1, 0):
    for first_mode in (0, 1, 2):
        idx = mi.IndexBipartite(d, n, 1, None)
        idx.SetBaseData(base); idx.SetProjectionGraph(3, off, nbrs); idx.set_option("hub_mode", 1)
        idx.set_option("calibrate", cal); idx.set_option("visited_mode", first_mode)
        for rep in range(3):
            r = idx.SearchRoarGraphBatch(q, 10, 20)
            print(f"calibrate {cal} visited_mode {first_mode} rep {rep}: cmps {r['cmps'].mean():.1f} (oracle {o['cmps'].mean():.1f}) scanned {r['scanned'].mean():.1f} "
                  f"ids equal {np.array_equal(r['ids'], o['ids'])} last mode {idx.info('last_visited_mode')} screen {idx.info('last_screen')} calibrated {idx.info('last_plan_calibrated')}")
        idx.close()
