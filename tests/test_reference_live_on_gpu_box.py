"""The link between the checker the `-m gpu` tests use (oracle/liboracle.so, the C restatement) and the reference's own
compiled code (oracle/_ref/libroar_ref.so: the reference's distance.h / neighbor.h / visited_list_pool.h / util.h built
where they lie, which travels to the GPU box as a built file) -- run ON the GPU box, in the same session as the parity
tests, so that what pins the HIP path there is the reference's code on that machine and not only fixtures made elsewhere.
These need no GPU; they carry the gpu marker because the GPU box only runs `-m gpu`."""
import os

import pytest

import oracle
from tests import test_oracle_pinning as pin

pytestmark = pytest.mark.gpu

REF_SO = os.path.join(os.path.dirname(os.path.abspath(oracle.__file__)), "_ref", "libroar_ref.so")


def _avx512():
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        return False
    return "avx512f" in flags and "avx512dq" in flags


def _require_ref():
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref/libroar_ref.so did not travel to this box")
    if not _avx512():
        pytest.skip("host CPU lacks AVX-512 F/DQ: the reference's distance.h cannot run here")
    assert oracle.have_ref(), "oracle/_ref/libroar_ref.so is present and the CPU has AVX-512, but it does not load"


def test_restatement_equals_reference_components_on_this_box(golden_dir):
    _require_ref()
    pin.test_port_equals_reference_components_live.__wrapped__(golden_dir) if hasattr(
        pin.test_port_equals_reference_components_live, "__wrapped__") else pin.test_port_equals_reference_components_live(golden_dir)


def test_bipartite_restatement_equals_reference_components_on_this_box():
    _require_ref()
    pin.test_bipartite_port_equals_reference_components_live()


def test_hip_search_equals_reference_components_directly(built_lib, golden_dir):
    """No restatement in between: the HIP path against a search run through the reference's compiled queue / visited pool /
    distance objects (impl="ref") on this box -- random graph, both metrics, short and long pools."""
    import numpy as np

    from mysteryann_amd import index as mi

    _require_ref()
    rng = np.random.default_rng(77)
    n, d = 4000, 200
    base = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((40, d)).astype(np.float32)
    deg = rng.integers(1, 40, n)
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum(deg)
    nbrs = rng.integers(0, n, int(off[-1])).astype(np.uint32)
    for metric, m in ((mi.L2, 0), (mi.INNER_PRODUCT, 1)):
        idx = mi.IndexBipartite(d, n, metric, None)
        idx.SetBaseData(base)
        idx.SetProjectionGraph(5, off, nbrs)
        idx.set_option("visited_mode", 1)  # the exact on-chip form: cmps is the reference's too (AUTO may take the forgetful one)
        for L in (10, 64, 300):
            r = idx.SearchRoarGraphBatch(q, 10, L)
            o = oracle.search_batch(base, off, nbrs, 5, q, 10, L, m, threads=2, impl="ref")
            for key in ("ids", "dists", "hops", "cmps"):
                assert np.array_equal(r[key], o[key]), (metric, L, key)
