"""Condense scripts/profile_r5.sh's capture into <out>/<tag>_search_kernel_summary.json (+ <tag>_kernel_stats.csv): the search kernel
one launch at a time (rocprofv3 average launch duration against HIP events) and the batch-stream regime `value` is measured in (the
stream's dispatches live for the whole timed loop: their duration / steps against the device-clock span / steps the bench reports)."""
import csv, glob, json, os, sys

src, tag = sys.argv[1], sys.argv[2]
KERNEL = "roar_search_v2_kernel"


def newest(pattern):
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


def launches(d):
    rows = [r for r in csv.DictReader(open(newest(os.path.join(src, d, "**/*_kernel_trace.csv"))))
            if KERNEL in r["Kernel_Name"] and "build" not in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]


bench = json.load(open(os.path.join(src, "bench.json")))
ts = json.load(open(os.path.join(src, "trace_serial.json")))
tf = json.load(open(os.path.join(src, "trace_stream.json")))
steps = ts["steps"]
ser = launches("trace_serial")[-steps:]
ser_d = [e - s for s, e, _ in ser]
st_all = launches("trace_stream")
# the stream's dispatches: the batch-stream instantiation (template argument SRC = 2), last loops of the process
stream_rows = [x for x in st_all if ", 2, " in x[2] or "Li2E" in x[2]]
# the timed loops run in the order: replayed / serial / launches in flight (per-launch kernels), then the stream (warm-up, timed), then nothing else
last = stream_rows[-2:] if len(stream_rows) >= 2 else stream_rows
summary = {
    "commands": {"bench": "python3 bench.py",
                 "one_launch_at_a_time": f"rocprofv3 --kernel-trace --stats -- python3 bench.py --steps {steps} --warmup 5 --in-flight 1 --no-stream --L {bench['config']['L_pq']} ...",
                 "stream": f"rocprofv3 --kernel-trace --stats -- python3 bench.py --steps {steps} --warmup 5 --L {bench['config']['L_pq']} ... (value through one batch stream)"},
    "one_launch_at_a_time": {
        "avg_launch_ms_rocprof": sum(ser_d) / len(ser_d) / 1e6,
        "launch_durations_ns_rocprof": ser_d,
        "kernel_ms_hip_events_same_process": ts["roofline"]["one_launch_at_a_time"]["kernel_ms"],
        "kernel_ms_hip_events_bench_run": bench["roofline"]["one_launch_at_a_time"]["kernel_ms"],
        "traffic_bytes_per_launch_pmc_bench_run": bench["roofline"]["one_launch_at_a_time"].get("traffic"),
        "frac_bench_run": bench["roofline"]["one_launch_at_a_time"].get("frac")},
    "batch_stream": {
        "dispatches_of_the_stream_kernel_in_the_traced_process": len(stream_rows),
        "last_dispatches_ms_rocprof": [(e - s) / 1e6 for s, e, _ in last],
        "note": "a dispatch of the batch-stream form lives from the first submit that finds fewer workgroups alive than the chip holds to the "
                "moment its last workgroup finds nothing published left: the timed loop's K steps are served by the last one or two of them",
        # the warm-up steps and the timed steps are separated by a wait that lets every workgroup exit, so the LAST dispatch is the timed
        # loop's alone (the one before it served the warm-up)
        "ms_per_step_rocprof": (last[-1][1] - last[-1][0]) / 1e6 / steps if last else None,
        "ms_per_step_rocprof_is": "duration of the LAST dispatch of the stream kernel (the timed loop's steps; the one before it served the warm-up) / steps",
        "kernel_ms_device_clock_same_process": tf["roofline"]["kernel_ms"], "regime_same_process": tf["config"]["value_regime"],
        "kernel_ms_device_clock_bench_run": bench["roofline"]["kernel_ms"], "regime_bench_run": bench["config"]["value_regime"],
        "traffic_bytes_per_launch_pmc_bench_run": bench["roofline"].get("traffic"), "frac_bench_run": bench["roofline"].get("frac")},
    "bench_line": bench,
}
json.dump(summary, open(os.path.join(src, f"{tag}_search_kernel_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "bench_line"}, indent=1))
# kernel stats of both traced processes
for d in ("trace_serial", "trace_stream"):
    try:
        f = newest(os.path.join(src, d, "**/*_kernel_stats.csv"))
        rows = list(csv.DictReader(open(f)))
        with open(os.path.join(src, f"{tag}_kernel_stats_{d}.csv"), "w") as o:
            w = csv.DictWriter(o, fieldnames=rows[0].keys())
            w.writeheader()
            for r in rows[:12]:
                r = dict(r)
                r["Name"] = r["Name"][:160]
                w.writerow(r)
    except Exception as e:  # noqa: BLE001
        print("no kernel stats for", d, e)
