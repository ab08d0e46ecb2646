import json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from mysteryann_amd import index as mi, synth
Y = json.load(open("tests/golden/builder_yardstick_1000000.json"))
n, nq = Y["n"], Y["nq"]
base, q, learn = synth.t2i_numpy(n, nq, 200, seed=Y["seed"], n_learn=Y["n_learn"])
idx = mi.IndexBipartite(200, n, mi.INNER_PRODUCT, None); idx.SetBaseData(base)
lists, _ = idx.ExactKnnDevice(torch.from_numpy(learn).cuda(), 100, want_dists=False)
p = mi.Parameters(); p.Set("M_sq", 100); p.Set("M_pjbp", 35); p.Set("L_pjpq", 500)
import time; t0 = time.time(); idx.BuildRoarGraphDevice(lists, p); torch.cuda.synchronize(); print("device build s", time.time() - t0)
print("ep", idx.info("ep"), Y["ep"], "avg", idx.info("nedges") / n, Y["avg_degree"], "max", idx.info("max_degree"), Y["max_degree"])
qd = torch.from_numpy(q).cuda()
gt = idx.ExactKnnDevice(qd, 10, want_dists=False)[0].cpu().numpy().view(np.uint32)
idx.set_option("visited_mode", 0)
for L in (20, 50, 100):
    r = idx.SearchRoarGraphBatch(q, 10, L)
    print(L, "device recall", mi.ComputeRecall(r["ids"], gt, 10), "cmps", r["cmps"].mean(), "yardstick", Y["curve"][str(L)])
