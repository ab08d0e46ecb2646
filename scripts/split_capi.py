#!/usr/bin/env python3
"""One-off (round 5): cut csrc/roar_capi.cpp (3 203 lines) into translation units by line ranges.  Kept for the record of what
moved where; the output files are the source of truth afterwards."""
import os
import sys

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mysteryann_amd", "csrc")
src = open(os.path.join(HERE, "roar_capi.cpp.orig")).read().split("\n")


def L(a, b):  # 1-indexed inclusive
    return src[a - 1:b]


def strip_defaults(lines):
    out = []
    for ln in lines:
        ln = ln.replace("bool bipartite = false, const PlanPick *pick = nullptr) {", "bool bipartite, const PlanPick *pick) {")
        out.append(ln)
    return out


HEAD = """// {name} -- {what}
// (round 5: roar_capi.cpp was one 3 203-line translation unit; it is now cut by concern.  Shared declarations: roar_internal.h.)
#include "roar_internal.h"

using namespace roar_impl;
"""


def emit(name, what, pieces):
    out = [HEAD.format(name=name, what=what)]
    for kind, lines in pieces:
        if kind == "impl":
            out.append("namespace roar_impl {\n")
            out.extend(lines)
            out.append("}  // namespace roar_impl\n")
        elif kind == "api":
            out.append('extern "C" {\n')
            out.extend(lines)
            out.append('}  // extern "C"\n')
        else:
            out.extend(lines)
    open(os.path.join(HERE, name), "w").write("\n".join(out))


# roar_capi.cpp: errors, allocation hook, the index object's buffers in HBM, lifecycle, base rows, graph files
emit("roar_capi.cpp", "C ABI (include/roargraph_amd.h): the index object -- errors, device buffers, base rows, graph files, visited pool.  "
     "Host-only code built with hipcc.  There is deliberately no CPU compute path in this library.",
     [("impl", L(93, 104) + L(112, 126)),
      ("impl", L(290, 486) + L(489, 590)),
      ("api", L(1023, 1320) + L(1737, 1745))])

emit("roar_plan.cpp", "launch plans of the SearchRoarGraph kernels: LDS carve-up, kernel form, grid (plan_search), and the calibration that "
     "measures what no option fixes on the loaded index (calibrate_plan, resolve_plan)",
     [("impl", strip_defaults(L(611, 919))),
      ("impl", L(1764, 1808) + L(1810, 2053))])

emit("roar_launch.cpp", "the device-buffer search call (roar_search_batch_device): plan, launch cell, kernel launch, ordering of launches that share "
     "per-workgroup lists in HBM",
     [("impl", L(2057, 2122)),
      ("api", L(1753, 1760))])

emit("roar_host.cpp", "the host-buffer search calls: query staging / packing behind a running launch, result fetch, combining of concurrent "
     "one-query calls, replicas (roar_clone_to_device, roar_search_batch_multi), pinned host memory",
     [("impl", L(921, 1017)),
      ("impl", L(2127, 2207)),
      ("impl", L(2227, 2279)),
      ("api", L(2210, 2224) + L(2282, 2429))])

emit("roar_buildapi.cpp", "graph builders behind the C ABI: roar_build_graph (bootstrap), roar_build_roargraph(_device) (the reference's BuildRoarGraph)",
     [("api", L(1322, 1736))])

emit("roar_bipartite.cpp", "the bipartite two-hop variant (SURVEY 8(f) row 4) behind the C ABI",
     [("impl", L(2435, 2460)),
      ("api", L(2463, 2684))])

emit("roar_knnapi.cpp", "the exact kNN generator behind the C ABI (roar_exact_knn(_device)): bf16 MFMA screen + f32 re-rank, f32 MFMA form",
     [("impl", L(2793, 2925)),
      ("api", L(2928, 2995))])

emit("roar_misc.cpp", "small entry points: roar_distances, the screen's known-answer surface, top-k merge, id rebase, queue trace, allocation-failure hook, "
     "phase timers, traffic counters, options and info keys",
     [("api", L(2686, 2790) + L(2997, 3201))])

# the header's raw material, for hand assembly
open(os.path.join(HERE, "roar_internal.h.parts"), "w").write("\n".join(
    ["// ---- includes + kernel launcher declarations"] + L(4, 89) +
    ["// ---- macros"] + L(105, 111) + L(127, 130) +
    ["// ---- struct"] + L(134, 269) +
    ["// ---- DeviceRestore"] + L(273, 288) +
    ["// ---- consts"] + L(487, 488) +
    ["// ---- plan types"] + L(592, 609)))
print("ok")
