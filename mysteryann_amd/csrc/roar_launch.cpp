// roar_launch.cpp -- the device-buffer search call (roar_search_batch_device): plan, launch cell, kernel launch, ordering of launches that share per-workgroup lists in HBM
// (round 5: roar_capi.cpp was one 3 203-line translation unit; it is now cut by concern.  Shared declarations: roar_internal.h.)
#include "roar_internal.h"

using namespace roar_impl;

namespace roar_impl {

// ready_dev: see RoarSearchParams::ready (the host-buffer path with a pageable query block)
int search_device(roar_index *idx, const float *queries_dev, uint64_t nq, uint32_t qstride_floats, uint32_t k, uint32_t L_pq,
                  uint32_t *ids_dev, float *dists_dev, uint32_t *cmps_dev, uint32_t *hops_dev, uint32_t *scanned_dev,
                  uint32_t *short_count_dev, void *hip_stream, const uint32_t *ready_dev, const LaunchPlan *planned) {
    if (!idx || !ids_dev || !dists_dev || (nq && !queries_dev)) return fail(ROAR_ERR_ARG, "bad search arguments");
    if (nq == 0) return ROAR_OK;
    if (qstride_floats < idx->dim_pad || (qstride_floats & 3))
        return fail(ROAR_ERR_ARG, "device queries need a stride >= %u floats that is a multiple of 4", idx->dim_pad);
    int e = use_device(idx);
    if (e) return e;
    hipStream_t st = (hipStream_t)hip_stream;
    LaunchPlan plan;
    if (planned) plan = *planned;
    else if ((e = resolve_plan(idx, queries_dev, nq, qstride_floats, k, L_pq, st, &plan, nullptr))) return e;
    RoarSearchParams &p = plan.p;
    p.queries = queries_dev;
    p.qstride = qstride_floats;
    p.out_ids = ids_dev;
    p.out_dists = dists_dev;
    p.out_cmps = cmps_dev;
    p.out_hops = hops_dev;
    p.out_scanned = scanned_dev;
    p.short_count = short_count_dev;
    // (the counting instantiation exists for the one-wave form without hub bitmap / phase timers / host gate)
    if (idx->traffic_dev && nq <= idx->traffic_cap && !idx->building) {
        if (ready_dev || plan.sel.form != ROAR_FORM_SEARCH || plan.sel.waves != 1 || plan.sel.prof)
            return fail(ROAR_ERR_ARG, "traffic counters are set, but this launch (several waves per query / phase timers / pageable host "
                                      "queries) has no counting form: option search_kernel = 1, device or pinned queries");
        p.out_traffic = idx->traffic_dev;
        plan.sel.stats = 1;
    }
    if (ready_dev) {  // (search_host asked gate_allowed() first)
        if (plan.sel.form != ROAR_FORM_SEARCH || plan.sel.waves != 1 || plan.sel.hub || plan.sel.prof)
            return fail(ROAR_ERR_ARG, "internal: this kernel form cannot wait for the packing thread");
        p.ready = ready_dev;
        plan.sel.gated = 1;
        plan.sel.ann = 0;  // (the gated form reads the screened rows' floats from the copy: no instantiation with the edge floats)
        p.adj_meta = nullptr;
        if (plan.sel.protect) {  // (the gated form has no instantiation that keeps pool members' tags: the plain forgetful table)
            plan.sel.protect = 0;
            p.vis_sets_log2 &= ~0x20000000u;
            idx->last_tag_protect = 0;
        }
    }
    // Launches of one index on different streams may overlap (each has its own launch cell) -- except where they would
    // share per-workgroup state in HBM: the EXACT-mode tag lists / the spill lists are indexed by workgroup, so such a
    // launch waits for the previous one.
    if (idx->ev_valid && idx->last_stream != st && (p.tags != nullptr || idx->last_used_tags))
        HIP_TRY(hipStreamWaitEvent(st, idx->ev1, 0));
    idx->last_stream = st;
    idx->last_used_tags = plan.sel.visited_mode == ROAR_VISITED_EXACT || p.tags != nullptr;
    const bool timed = idx->launch_events != 0 || p.tags != nullptr;  // (launches that share tag lists are ordered by ev1)
    if (timed) HIP_TRY(hipEventRecord(idx->ev0, st));
    if ((e = launch_planned(idx, plan, st))) return e;
    if (timed) {
        HIP_TRY(hipEventRecord(idx->ev1, st));
        idx->ev_valid = true;
    }
    return ROAR_OK;
}

// the launch itself: a cell of the ring, the kernel
int launch_planned(roar_index *idx, LaunchPlan &plan, hipStream_t st) {
    // (the cell's stamp: which use of the cell this launch is, from 1; it expects to find its predecessor's -- RoarSearchParams::counter)
    const uint32_t seq = idx->launch_seq.fetch_add(1);
    plan.p.counter = idx->ring + 4 * (size_t)(seq % LAUNCH_RING);
    plan.p.stamp = idx->ring + 4 * (size_t)LAUNCH_RING + (seq % LAUNCH_RING);  // (the stamps: an array of their own behind the cells)
    plan.p.cell_prev = seq / LAUNCH_RING;
    plan.p.cell_stamp = seq / LAUNCH_RING + 1u;
    int rc = roar_launch_search(&plan.p, &plan.sel, plan.grid, st);
    if (rc != 0) {
        // the launch that was to use the cell never ran: its stamp is rolled forward in stream order, as the kernel's last workgroup would
        // have left it, or the cell would refuse its next user LAUNCH_RING launches from now (the cell itself is untouched: zero)
        (void)hipGetLastError();
        const int e = fail(ROAR_ERR_DEVICE, "search kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(const_cast<uint32_t *>(plan.p.stamp)), (int)plan.p.cell_stamp, 1, st) != hipSuccess) (void)hipGetLastError();
        return e;
    }
    return ROAR_OK;
}
}  // namespace roar_impl

extern "C" {

int roar_search_batch_device(roar_index *idx, const float *queries_dev, uint64_t nq, uint32_t qstride_floats,
                             uint32_t k, uint32_t L_pq, uint32_t *ids_dev, float *dists_dev, uint32_t *cmps_dev,
                             uint32_t *hops_dev, uint32_t *scanned_dev, uint32_t *short_count_dev,
                             void *hip_stream) {
    DeviceRestore keep_device_;
    return search_device(idx, queries_dev, nq, qstride_floats, k, L_pq, ids_dev, dists_dev, cmps_dev, hops_dev, scanned_dev,
                         short_count_dev, hip_stream, nullptr);
}
}  // extern "C"
