"""ctypes binding of the C ABI (include/roargraph_amd.h) -- the only way Python reaches the
product.  There is no Python / CPU implementation of the search behind it: if the HIP library is
missing or no GPU is present every compute call raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# (ROAR_AMD_LIB: another build of the same library, e.g. one compiled with different register budgets, for A/B measurements)
LIB_PATH = os.environ.get("ROAR_AMD_LIB") or os.path.join(_CSRC, "libroargraph_amd.so")

METRIC_L2, METRIC_IP, METRIC_COSINE = 0, 1, 4
VISITED_EXACT, VISITED_LDS, VISITED_AUTO = 0, 1, 2
OK, ERR_ARG, ERR_IO, ERR_FORMAT, ERR_DEVICE, ERR_SHORT = 0, -1, -2, -3, -4, -5

# every symbol include/roargraph_amd.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "roar_create", "roar_destroy", "roar_last_error", "roar_load_base_fbin", "roar_set_base",
    "roar_set_base_device", "roar_load_graph_file", "roar_set_graph_csr", "roar_save_graph_file",
    "roar_get_graph_csr", "roar_build_graph", "roar_build_roargraph", "roar_search_batch",
    "roar_search_batch_device", "roar_distances", "roar_exact_knn_device", "roar_exact_knn",
    "roar_merge_topk_device", "roar_set_option", "roar_get_info", "roar_init_visited_pool",
    "roar_load_bipartite_file", "roar_save_bipartite_file", "roar_set_bipartite_csr", "roar_get_bipartite_csr",
    "roar_build_bipartite", "roar_bipartite_entry_ids", "roar_search_bipartite_batch",
    "roar_search_bipartite_batch_device", "roar_last_profile", "roar_clone_to_device", "roar_search_batch_multi",
    "roar_host_alloc", "roar_host_free", "roar_queue_trace", "roar_build_roargraph_device",
    "roar_rebase_ids_device", "roar_sharded_create", "roar_sharded_destroy", "roar_sharded_shard",
    "roar_sharded_set_id_base", "roar_search_batch_sharded", "roar_sharded_info", "roar_host_register",
    "roar_host_unregister", "roar_debug_fail_device_alloc_at", "roar_debug_screen_keeps",
    "roar_set_traffic_counters_device", "roar_search_stream_open", "roar_search_stream_submit", "roar_search_stream_wait",
    "roar_search_stream_times", "roar_search_stream_info", "roar_search_stream_close",
]


class BuildParams(C.Structure):
    """struct roar_build_params (include/roargraph_amd.h)"""
    _fields_ = [("M", C.c_uint32), ("base_cap", C.c_uint32), ("max_degree", C.c_uint32), ("L_build", C.c_uint32),
                ("max_batch", C.c_uint64), ("learn_dev", C.c_void_p), ("n_learn", C.c_uint64),
                ("learn_stride_floats", C.c_uint32), ("M_sq", C.c_uint32), ("M_pj", C.c_uint32),
                ("L_learn", C.c_uint32), ("rounds", C.c_uint32), ("phase1_l2", C.c_uint32), ("n_knn", C.c_uint64),
                ("learn_knn_dev", C.c_void_p), ("supply_k", C.c_uint32), ("supply_L", C.c_uint32),
                ("L_learn_last", C.c_uint32), ("phase1_cand", C.c_uint32), ("supply_cand", C.c_uint32),
                ("proj_mode", C.c_uint32), ("supply_mode", C.c_uint32), ("no_skeleton", C.c_uint32),
                ("trim_min_proj", C.c_uint32), ("trim_keep_base", C.c_uint32), ("row_keep", C.c_uint32)]


class RoarError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def build_library(force=False):
    """hipcc --offload-arch=gfx950 build of csrc/ (cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in sorted(os.listdir(_CSRC)) if f.endswith((".hip", ".h")) or (f.endswith(".cpp") and not f.startswith("harness_"))]
    srcs.append(os.path.join(_CSRC, "..", "..", "include", "roargraph_amd.h"))
    stale = force or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-j8", "-C", _CSRC, "lib"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        try:  # in-tree build (hipcc cross-compiles gfx950 without a GPU)
            build_library()
        except (OSError, subprocess.CalledProcessError) as e:
            raise RoarError(ERR_DEVICE, f"libroargraph_amd.so is not built and cannot be built here ({e}); "
                                        "there is no fallback implementation") from e
    # One HIP runtime per process: torch bundles its own libamdhip64.so.7 and the system one has the
    # same SONAME, so whichever is mapped first serves both.  torch refuses to see the GPU on the
    # system runtime, therefore let torch map its copy before this library is loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    l = C.CDLL(LIB_PATH)
    vp, u32p, u64p, fp = C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_float)
    l.roar_last_error.restype = C.c_char_p
    l.roar_create.argtypes = [C.POINTER(vp), C.c_uint32, C.c_int, C.c_int]
    l.roar_destroy.argtypes = [vp]
    l.roar_destroy.restype = None
    l.roar_load_base_fbin.argtypes = [vp, C.c_char_p]
    l.roar_set_base.argtypes = [vp, fp, C.c_uint64, C.c_uint32]
    l.roar_set_base_device.argtypes = [vp, vp, C.c_uint64, C.c_uint32]
    l.roar_load_graph_file.argtypes = [vp, C.c_char_p]
    l.roar_set_graph_csr.argtypes = [vp, C.c_uint32, C.c_uint64, u64p, u32p]
    l.roar_save_graph_file.argtypes = [vp, C.c_char_p]
    l.roar_get_graph_csr.argtypes = [vp, u32p, u64p, u64p, u64p, u32p]
    l.roar_build_graph.argtypes = [vp, C.POINTER(BuildParams)]
    l.roar_build_roargraph.argtypes = [vp, u32p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    l.roar_build_roargraph_device.argtypes = [vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    l.roar_search_batch.argtypes = [vp, fp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, u32p, fp, u32p, u32p, u32p]
    l.roar_search_batch_device.argtypes = [vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp,
                                           vp, vp]
    l.roar_distances.argtypes = [vp, fp, C.c_uint64, C.c_uint32, u32p, C.c_uint32, fp]
    l.roar_exact_knn_device.argtypes = [vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, vp, vp, vp]
    l.roar_exact_knn.argtypes = [vp, fp, C.c_uint64, C.c_uint32, C.c_uint32, u32p, fp]
    l.roar_merge_topk_device.argtypes = [C.c_int, C.c_uint32, C.c_uint64, C.c_uint32, vp, vp, vp, vp, vp]
    l.roar_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    l.roar_get_info.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64)]
    l.roar_init_visited_pool.argtypes = [vp, C.c_uint32]
    l.roar_last_profile.argtypes = [vp, u64p, C.c_uint64]
    l.roar_clone_to_device.argtypes = [vp, C.c_int, C.POINTER(vp)]
    l.roar_search_batch_multi.argtypes = [C.POINTER(vp), C.c_uint32, fp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, u32p, fp,
                                          u32p, u32p, u32p]
    l.roar_queue_trace.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8), u32p, fp, C.c_int, u32p]
    l.roar_host_alloc.argtypes = [C.c_uint64]
    l.roar_host_alloc.restype = vp
    l.roar_host_free.argtypes = [vp]
    l.roar_host_free.restype = None
    l.roar_load_bipartite_file.argtypes = [vp, C.c_char_p]
    l.roar_save_bipartite_file.argtypes = [vp, C.c_char_p]
    l.roar_set_bipartite_csr.argtypes = [vp, C.c_uint64, u64p, u32p]
    l.roar_get_bipartite_csr.argtypes = [vp, u64p, u64p, u64p, u32p]
    l.roar_build_bipartite.argtypes = [vp, u32p, C.c_uint64, C.c_uint32, C.c_uint32]
    l.roar_bipartite_entry_ids.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, u32p]
    l.roar_search_bipartite_batch.argtypes = [vp, fp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, u32p, C.c_uint32,
                                              u32p, fp, u32p, u32p]
    l.roar_search_bipartite_batch_device.argtypes = [vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, vp,
                                                     C.c_uint32, vp, vp, vp, vp, vp, vp, vp]
    l.roar_debug_fail_device_alloc_at.argtypes = [C.c_int64]
    l.roar_debug_screen_keeps.argtypes = [vp, fp, C.c_uint64, C.c_uint32, u32p, fp, C.c_uint32, C.POINTER(C.c_uint8)]
    l.roar_set_traffic_counters_device.argtypes = [vp, vp, C.c_uint64]
    l.roar_search_stream_open.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    l.roar_search_stream_submit.argtypes = [vp, vp, C.c_uint64, C.c_uint32, vp, vp, vp, vp, vp, u64p]
    l.roar_search_stream_wait.argtypes = [vp, C.c_uint64, u32p]
    l.roar_search_stream_times.argtypes = [vp, C.c_uint64, u64p, u64p]
    l.roar_search_stream_info.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64)]
    l.roar_search_stream_close.argtypes = [vp]
    l.roar_host_register.argtypes = [vp, C.c_uint64]
    l.roar_host_unregister.argtypes = [vp]
    l.roar_rebase_ids_device.argtypes = [C.c_int, vp, C.c_uint64, C.c_uint32, vp]
    l.roar_sharded_create.argtypes = [C.POINTER(vp), C.c_uint32, C.c_int, C.POINTER(C.c_int), C.c_uint32]
    l.roar_sharded_destroy.argtypes = [vp]
    l.roar_sharded_destroy.restype = None
    l.roar_sharded_shard.argtypes = [vp, C.c_uint32]
    l.roar_sharded_shard.restype = vp
    l.roar_sharded_set_id_base.argtypes = [vp, C.c_uint32, C.c_uint64]
    l.roar_search_batch_sharded.argtypes = [vp, fp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, u32p, fp, u32p, u32p]
    l.roar_sharded_info.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64)]
    for s in SYMBOLS:
        if s not in ("roar_destroy", "roar_last_error", "roar_host_alloc", "roar_host_free", "roar_sharded_destroy",
                     "roar_sharded_shard"):
            getattr(l, s).restype = C.c_int
    _lib = l
    return l


def check(rc):
    if rc != OK:
        raise RoarError(rc, lib().roar_last_error().decode("utf-8", "replace"))


def f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def u64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))
