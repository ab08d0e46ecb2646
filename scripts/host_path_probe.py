import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from _graph import t2i_index
from mysteryann_amd import capi
idx, base, queries = t2i_index(10_000_000, 10000, "reference", 0, "/tmp/r2_graph.index")
lib = capi.lib()
hq_all = queries.cpu().numpy()
for nq in (64, 1250, 10000):
    hq = np.ascontiguousarray(hq_all[:nq])
    ptr = lib.roar_host_alloc(hq.nbytes)
    pin = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=hq.shape); pin[:] = hq
    ids = np.zeros((nq, 10), np.uint32); dists = np.zeros((nq, 10), np.float32)
    def call(q):
        capi.check(lib.roar_search_batch(idx._h, capi.f32p(q), nq, 200, 10, 55, capi.u32p(ids), capi.f32p(dists), None, None, None))
    for name, q in (("pinned", pin), ("pageable", hq)):
        for _ in range(5): call(q)
        t0 = time.perf_counter()
        for _ in range(40): call(q)
        dt = (time.perf_counter() - t0) / 40
        print(f"nq={nq} {name}: {dt * 1e3:.3f} ms per call, kernel {idx.info('last_kernel_ns') / 1e6:.3f} ms", flush=True)
    lib.roar_host_free(ptr)
