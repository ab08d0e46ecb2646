"""Why does the yardstick test see cmps == scanned at L_pq = 20 in EXACT mode?  Its exact flow, with variations."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mysteryann_amd import index as mi, synth
Y = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "builder_yardstick_200000.json")))
n, nq = Y["n"], Y["nq"]
base, q, learn = synth.t2i_numpy(n, nq, 200, seed=Y["seed"], n_learn=Y["n_learn"])
for variant in ("test flow", "no gt call", "calibrate 0 first", "device queries", "knn f32"):
    idx = mi.IndexBipartite(200, n, mi.INNER_PRODUCT, None)
    idx.SetBaseData(base)
    if variant == "knn f32":
        idx.set_option("knn_mode", 1)
    lists, _ = idx.ExactKnnDevice(torch.from_numpy(learn).cuda(), 100, want_dists=False)
    p = mi.Parameters()
    p.Set("M_sq", Y["M_sq"]), p.Set("M_pjbp", Y["M_pjbp"]), p.Set("L_pjpq", Y["L_pjpq"])
    idx.BuildRoarGraphDevice(lists, p)
    qd = torch.from_numpy(q).cuda()
    if variant != "no gt call":
        gt = idx.ExactKnnDevice(qd, 10, want_dists=False)[0].cpu().numpy().view(np.uint32)
    idx.set_option("visited_mode", 0)
    if variant == "calibrate 0 first":
        idx.set_option("calibrate", 0)
    for rep, L in enumerate((20, 20, 50, 20)):
        if variant == "device queries":
            o = idx.SearchRoarGraphBatchDevice(qd, 10, L)
            torch.cuda.synchronize()
            r = {k_: v.cpu().numpy() for k_, v in o.items()}
        else:
            r = idx.SearchRoarGraphBatch(q, 10, L)
        print(f"{variant}: rep {rep} L {L}: cmps {r['cmps'].mean():.1f} hops {r['hops'].mean():.2f} scanned {r['scanned'].mean():.1f} mode {idx.info('last_visited_mode')} "
              f"screen {idx.info('last_screen')} calibrated {idx.info('last_plan_calibrated')} grid {idx.info('last_grid')}", flush=True)
        if variant == "calibrate 0 first":
            idx.set_option("calibrate", 1)
    idx.close()
