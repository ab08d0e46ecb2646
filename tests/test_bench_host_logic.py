"""Host-side logic of bench.py that runs without a GPU: the counter-file reader of the in-run HBM traffic measurement
and the query split of the strong-scaling launch."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _csv(path, rows):
    with open(path, "w") as f:
        f.write("Correlation_Id,Dispatch_Id,Agent_Id,Kernel_Name,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp\n")
        for i, (name, counter, value) in enumerate(rows):
            f.write(f"{i},{i},1,\"{name}\",{counter},{value},{1000 * i},{1000 * i + 500}\n")


def test_counter_per_launch_takes_the_last_launches_of_the_search_kernel(tmp_path):
    import bench

    search = "void (anonymous namespace)::roar_search_v2_kernel<1, true, 12, true, 1, true, false, false>(RoarSearchParams)"
    build = "void (anonymous namespace)::roar_build_search_v2_kernel<true, 12, true, true>(RoarSearchParams)"
    rows = [(build, "FETCH_SIZE", 7.0), (search, "FETCH_SIZE", 1000.0),  # builder search, warm-up launch
            ("knn_pass_kernel", "FETCH_SIZE", 5e6)]
    rows += [(search, "FETCH_SIZE", 100.0 + i) for i in range(4)]
    rows += [(search, "WRITE_SIZE", 1.0)]  # another counter's row is not this counter's launch
    p = tmp_path / "x_counter_collection.csv"
    _csv(p, rows)
    assert bench.counter_per_launch(str(p), "FETCH_SIZE", 4) == pytest.approx(101.5)
    assert bench.counter_per_launch(str(p), "FETCH_SIZE", 5) == pytest.approx((1000.0 + 100 + 101 + 102 + 103) / 5)
    with pytest.raises(RuntimeError):
        bench.counter_per_launch(str(p), "FETCH_SIZE", 6)


def test_strong_scaling_split_covers_the_batch_once():
    from mysteryann_amd import parallel

    for nq in (1, 7, 10_000):
        for g in (1, 2, 3, 8):
            spans = [parallel.shard_range(nq, g, r) for r in range(g)]
            assert spans[0][0] == 0 and spans[-1][1] == nq
            assert all(spans[r][1] == spans[r + 1][0] for r in range(g - 1))
            assert all(lo == nq * r // g and hi == nq * (r + 1) // g for r, (lo, hi) in enumerate(spans))  # roar_search_batch_multi's rule


def test_plain_multi_gpu_command_starts_the_ranks_as_a_child(monkeypatch):
    """`python bench.py --gpus 8 ...` as the driver types it: before torch is imported the script starts
    torch.distributed.run with one rank per GPU as a child process and returns its exit code; under a launcher
    (RANK set), at N = 1 and in the profiled child it goes on itself."""
    import subprocess

    import bench

    argv = ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    cmd = bench.launcher_command(argv, 8, 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    script = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[script + 1:] == argv  # the script's own arguments pass through unchanged, after the script
    # `--n` would be taken by the launcher's own parser as an abbreviation of --nnodes / --nproc-per-node: it is spelled out
    assert bench.launcher_command(["--gpus", "2", "--n", "1000"], 2, 1)[-4:] == ["--gpus", "2", "--base-points", "1000"]
    calls = []
    monkeypatch.setattr(subprocess, "call", lambda c, env=None: calls.append((c, env)) or 17)
    monkeypatch.delenv("RANK", raising=False)
    assert bench.self_launch(bench.parse_args(argv), argv) == 17
    (c, env), = calls
    assert c[-len(argv):] == argv and env["MASTER_ADDR"] == "127.0.0.1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert bench.self_launch(bench.parse_args(["--gpus", "1"]), ["--gpus", "1"]) is None
    monkeypatch.setenv("RANK", "3")
    assert bench.self_launch(bench.parse_args(argv), argv) is None
    assert len(calls) == 1
    assert 1024 < bench.free_port() < 65536


def test_plain_multi_gpu_command_runs_end_to_end_without_a_gpu():
    """The real thing on this CPU box: the child ranks start, find no GPU and say so; the parent hands their failure
    back as its own exit code instead of dying on the WORLD_SIZE check (round 2's behaviour)."""
    import subprocess

    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"),
                        "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
    assert r.returncode != 0
    assert "torch.distributed.run" in r.stderr and "needs an MI355X" in r.stderr
    assert "WORLD_SIZE is" not in r.stderr


def test_implemented_bytes_formula():
    """roofline.bytes_per_launch: screen rows + f32 rows + adjacency words + query read + result write."""
    import bench

    # 10 queries, 200-d padded, 256-byte screen rows, k = 10: 1000 coded rows, 150 f32 rows, 6400 adjacency words
    got = bench.implemented_bytes(1000, 150, 6400, 10, 200, 256, 10)
    assert got == 256 * 1000 + 800 * 150 + 4 * 6400 + (800 + 80) * 10
    # screen off: nothing coded, the reference's 4 D per distance evaluation
    assert bench.implemented_bytes(0, 2840, 0, 1, 200, 256, 10) == 800 * 2840 + 880


def test_kernel_trace_span_of_overlapping_launches(tmp_path):
    import bench

    search = "void (anonymous namespace)::roar_search_v2_kernel<1, true, 12, true, 1, true, false, false, false, false>(RoarSearchParams)"
    p = tmp_path / "x_kernel_trace.csv"
    with open(p, "w") as f:
        f.write("Kind,Agent_Id,Queue_Id,Kernel_Name,Start_Timestamp,End_Timestamp\n")
        f.write(f'KERNEL_DISPATCH,1,1,"knn_pass_kernel",0,100\n')
        f.write(f'KERNEL_DISPATCH,1,1,"{search}",1000,2000\n')  # warm-up, outside the last four
        for i in range(4):  # four launches overlapping two at a time: 3 ms each, started 1.5 ms apart
            f.write(f'KERNEL_DISPATCH,1,{i % 2},"{search}",{10_000_000 + 1_500_000 * i},{13_000_000 + 1_500_000 * i}\n')
    span, mean = bench.kernel_trace_span(str(p), 4)
    assert span == pytest.approx(7.5) and mean == pytest.approx(3.0)
    with pytest.raises(RuntimeError):
        bench.kernel_trace_span(str(p), 6)


def test_emulated_share_of_8_block():
    """config.emulated_share_of_8 (N = 1 runs): the projection arithmetic."""
    import bench

    # 10 000-query batch, 20 steps: one rank's 1 250-query share took 4.4 ms in all, six in flight; this run served 7.0 M QPS
    b = bench.share_of_8_block(10_000, 20, 4.4e-3, 8.6e-3, 6, 640, 7.0e6)
    assert b["queries_per_step"] == 1250 and b["launches_in_flight"] == 6 and b["workgroups_per_launch_cap"] == 640
    assert b["ms_per_step"] == pytest.approx(0.22) and b["ms_per_step_one_launch_at_a_time"] == pytest.approx(0.43)
    assert b["projected_value_8_gpus"] == pytest.approx(10_000 * 20 / 4.4e-3)
    assert b["projected_speedup_over_this_run"] == pytest.approx(10_000 * 20 / 4.4e-3 / 7.0e6)
    assert "projection" in b["note"] and b["regime"] == "6 launches in flight" and b["ms_per_step_batch_stream"] is None
    # the same share through one batch stream in 3.9 ms: the better regime is the one projected from, and it is named
    b = bench.share_of_8_block(10_000, 20, 4.4e-3, 8.6e-3, 6, 640, 7.0e6, 3.9e-3, {"kernel_launches_in_the_timed_region": 3})
    assert b["regime"].startswith("one batch stream") and b["ms_per_step"] == pytest.approx(0.195)
    assert b["ms_per_step_launches_in_flight"] == pytest.approx(0.22) and b["ms_per_step_batch_stream"] == pytest.approx(0.195)
    assert b["projected_speedup_over_this_run"] == pytest.approx(10_000 * 20 / 3.9e-3 / 7.0e6)
    # ... and a stream that is slower is reported but not projected from
    b = bench.share_of_8_block(10_000, 20, 4.4e-3, 8.6e-3, 6, 640, 7.0e6, 5.0e-3, None)
    assert b["regime"] == "6 launches in flight" and b["ms_per_step"] == pytest.approx(0.22)
