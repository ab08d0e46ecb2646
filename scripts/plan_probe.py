"""What the launch-plan calibration picks on the benchmark index, beside explicit settings (round 3's hand-tuned plan among them):
kernel ms (HIP events, best of 3) per (queries per launch, L_pq).
  ROAR_LOG_CALIBRATION=1 python scripts/plan_probe.py [--n N] [--nqs ...] [--L ...]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from _graph import t2i_index

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--graph-cache", default="/tmp/roar_graph.index")
ap.add_argument("--L", type=int, nargs="+", default=[50, 100, 500, 1000, 2000])
ap.add_argument("--nqs", type=int, nargs="+", default=[1250, 2500, 10000])
args = ap.parse_args()
k = 10
idx, base, queries = t2i_index(args.n, 10000, "reference", 0, args.graph_cache)


def ms(q, L, reps=3):
    best = 1e9
    for _ in range(reps):
        idx.SearchRoarGraphBatchDevice(q, k, L)
        torch.cuda.synchronize()
        best = min(best, idx.info("last_kernel_ns") / 1e6)
    return best


def setopts(**kw):
    for key in ("screen", "visited_mode", "vis_entries", "calibrate"):
        idx.set_option(key, kw.get(key, {"screen": 0, "visited_mode": 2, "vis_entries": 0, "calibrate": 1}[key]))


for nq in args.nqs:
    q = queries[:nq].contiguous()
    for L in args.L:
        setopts()
        t_auto = ms(q, L, 4)
        desc = (f"screen {idx.info('last_screen')} mode {idx.info('last_visited_mode')} tags {idx.info('last_vis_entries')} "
                f"calibrated {idx.info('last_plan_calibrated')}")
        setopts(calibrate=0)
        t_def = ms(q, L)
        # round 3's hand-tuned plan for this graph: screen on, forgetful table of 4 096 tags below L_pq 300, 2 048 to 1 550, HBM tags beyond
        if L < 1550:
            setopts(calibrate=0, screen=2, visited_mode=3, vis_entries=4096 if L < 300 else 2048)
        else:
            setopts(calibrate=0, screen=2, visited_mode=0)
        t_r3 = ms(q, L)
        row = [f"nq {nq:5d} L_pq {L:4d}: auto {t_auto:8.3f} ms ({desc}) | defaults {t_def:8.3f} | round-3 plan {t_r3:8.3f}"]
        for ve in (2048, 4096, 8192):
            setopts(calibrate=0, screen=2, visited_mode=3, vis_entries=ve)
            row.append(f"{ve} tags {ms(q, L):8.3f}")
        setopts(calibrate=0, screen=1)
        row.append(f"screen off {ms(q, L):8.3f}")
        print(" | ".join(row), flush=True)
