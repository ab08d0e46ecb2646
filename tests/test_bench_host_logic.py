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
