"""Condense scripts/profile_r4.sh's capture into profiles/<tag>_search_kernel_summary.json (+ <tag>_kernel_stats.csv): the search
kernel in BOTH regimes -- one launch at a time (average launch duration, rocprofv3 against HIP events) and the timed loop's six
launches in flight (span of the launches / their number, rocprofv3 against HIP events) -- with the in-run PMC traffic of each."""
import csv, glob, json, os, sys

src, tag = sys.argv[1], sys.argv[2]
out = src
KERNEL = "roar_search_v2_kernel"


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)


def launches(d, steps):
    rows = [r for r in csv.DictReader(open(newest(os.path.join(src, d, "*/*_kernel_trace.csv"))))
            if KERNEL in r["Kernel_Name"] and "build" not in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # three timed loops of `steps` launches close the process (same batch replayed, one at a time, in flight): the last `steps` are the last loop
    return [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]


bench = json.load(open(os.path.join(src, "bench.json")))
ts = json.load(open(os.path.join(src, "trace_serial.json")))
tf = json.load(open(os.path.join(src, "trace_flight.json")))
steps = ts["steps"]
rf = bench["roofline"]
ser = launches("trace_serial", steps)[-steps:]
fl = launches("trace_flight", steps)[-steps:]
ser_d = [e - s for s, e in ser]
fl_span = (max(e for _, e in fl) - min(s for s, _ in fl)) / 1e6
summary = {
    "commands": {"bench": "python3 bench.py",
                 "one_launch_at_a_time": f"rocprofv3 --kernel-trace --stats -- python3 bench.py --steps {steps} --warmup 5 --in-flight 1 --L {bench['config']['L_pq']} ...",
                 "in_flight": f"rocprofv3 --kernel-trace --stats -- python3 bench.py --steps {steps} --warmup 5 --L {bench['config']['L_pq']} ... (six launches in flight)"},
    "kernel": "roar_search_v2_kernel<VIS_LDS, IP, NCH=12, TAIL, W=1, register pool> (D=200)",
    "one_launch_at_a_time": {
        "avg_launch_ms_rocprof": sum(ser_d) / len(ser_d) / 1e6,
        "launch_durations_ns_rocprof": ser_d,
        "kernel_ms_hip_events_same_process": ts["roofline"]["one_launch_at_a_time"]["kernel_ms"],
        "kernel_ms_hip_events_bench": rf["one_launch_at_a_time"]["kernel_ms"],
        "traffic_bytes_per_launch_pmc": rf["one_launch_at_a_time"].get("traffic"),
        "frac": rf["one_launch_at_a_time"]["frac"],
        "requested_bytes_frac_of_peak": rf["one_launch_at_a_time"].get("requested_bytes_frac_of_peak"),
    },
    "in_flight": {
        "launches_in_flight": bench["config"]["launches_in_flight"],
        "span_ms_rocprof": fl_span, "launches": len(fl), "span_ms_per_launch_rocprof": fl_span / len(fl),
        "mean_launch_duration_ms_rocprof": sum(e - s for s, e in fl) / len(fl) / 1e6,
        "kernel_ms_hip_events_same_process": tf["roofline"]["kernel_ms"],
        "ms_per_step_wall_same_process": tf["ms_per_step"],
        "kernel_ms_hip_events_bench": rf["kernel_ms"], "ms_per_step_wall_bench": bench["ms_per_step"],
        "traffic_bytes_per_launch_pmc": rf.get("traffic"), "frac": rf["frac"],
        "requested_bytes_frac_of_peak": rf.get("requested_bytes_frac_of_peak"),
    },
    "bytes_per_launch_implemented_algorithm": rf["bytes_per_launch"],
    "reference_bytes_per_launch": rf["reference_bytes_per_launch"],
    "hbm_traffic_bytes_per_launch_corrected": rf.get("traffic"),
    "correction": "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B) + WRITE_SIZE, measured by bench.py's own --pmc child passes",
    "bench_line": bench,
}
json.dump(summary, open(os.path.join(out, f"{tag}_search_kernel_summary.json"), "w"), indent=1)
stats = list(csv.DictReader(open(newest(os.path.join(src, "trace_serial/*/*_kernel_stats.csv")))))
with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps {steps} --warmup 5 --in-flight 1 --no-cpu-baseline --no-host-path --no-traffic --L {bench['config']['L_pq']}\n")
    f.write("# (whole process: data generation, the graph loaded from the file the bench run before it wrote, ground truth by the bf16-screened exact kNN; the L sweep is skipped with --L)\n")
    f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
    for r in stats[:14]:
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0][-90:] if "Cijk" not in name else "hipBLASLt GEMM (torch: synthetic data)"
        f.write(f"\"{name}\",{r['Calls']},{r['TotalDurationNs']},{r['AverageNs']},{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")
print(json.dumps({k: v for k, v in summary.items() if k != "bench_line"}, indent=1))
