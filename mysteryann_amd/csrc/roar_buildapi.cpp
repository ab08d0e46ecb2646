// roar_buildapi.cpp -- graph builders behind the C ABI: roar_build_graph (bootstrap), roar_build_roargraph(_device) (the reference's BuildRoarGraph)
// (round 5: roar_capi.cpp was one 3 203-line translation unit; it is now cut by concern.  Shared declarations: roar_internal.h.)
#include "roar_internal.h"

using namespace roar_impl;

extern "C" {

int roar_build_graph(roar_index *idx, const roar_build_params *bp) {
    DeviceRestore keep_device_;
    if (int busy_ = refuse_while_streaming(idx)) return busy_;
    if (!idx || !bp) return fail(ROAR_ERR_ARG, "bad arguments");
    if (!idx->base || idx->n < 2) return fail(ROAR_ERR_ARG, "build needs base vectors in HBM");
    const uint32_t M = bp->M, base_cap = bp->base_cap, max_degree = bp->max_degree, L_build = bp->L_build;
    const uint32_t C = bp->M_sq, Mp = bp->M_pj, L_learn = bp->L_learn, rounds = bp->rounds;
    const uint64_t n_learn = bp->learn_dev ? bp->n_learn : 0;
    if (M == 0 || M > 64 || base_cap < M || max_degree < base_cap || max_degree > 1023 || L_build < M)
        return fail(ROAR_ERR_ARG, "need 1 <= M <= 64, M <= base_cap <= max_degree <= 1023, L_build >= M");
    if (idx->n >= (1ull << 31)) return fail(ROAR_ERR_ARG, "more than 2^31 base points are not supported");
    if (n_learn && (bp->learn_stride_floats < idx->dim_pad || (bp->learn_stride_floats & 3) || C < 2 || C > 64 ||
                    Mp == 0 || L_learn < C))
        return fail(ROAR_ERR_ARG, "bad training-query arguments (need 2 <= M_sq <= 64 <= ..., L_learn >= M_sq)");
    int e = use_device(idx);
    if (e) return e;
    free_graph(idx);
    const uint64_t n = idx->n;
    const uint32_t S = round_up(max_degree + 1, 32);
    HIP_TRY(roar_dev_malloc((void **)&idx->adj, ((size_t)n * S + ADJ_SLACK_WORDS) * sizeof(uint32_t)));
    HIP_TRY(hipMemset(idx->adj, 0, ((size_t)n * S + ADJ_SLACK_WORDS) * sizeof(uint32_t)));
    idx->adj_stride = S;
    idx->gn = n;
    idx->max_deg = max_degree;
    uint64_t max_batch = bp->max_batch ? bp->max_batch : (1u << 19);
    max_batch = std::min<uint64_t>(max_batch, std::max<uint64_t>(n, std::max<uint64_t>(n_learn, bp->n_knn)));
    const uint64_t cap = std::max<uint64_t>(max_batch, 1025);
    const uint32_t C1 = bp->phase1_cand, C3 = bp->supply_cand;
    if (C1 > 63 || C3 > 63 || (C1 && C1 < M) || (C3 && C3 < bp->supply_k))
        return fail(ROAR_ERR_ARG, "phase1_cand / supply_cand must be 0 or in [M | supply_k, 63]");
    const uint32_t K = std::max(std::max(std::max(M, (n_learn || bp->learn_knn_dev) ? C : 0u), bp->supply_k + 1),
                                std::max(C1, C3));  // result row width
    if (bp->learn_knn_dev && (C < 2 || C > 64 || Mp == 0)) return fail(ROAR_ERR_ARG, "bad M_sq / M_pj for the kNN lists");
    uint32_t *res_ids = nullptr, *indeg = nullptr, *scratch = nullptr, *base_deg = nullptr;
    float *res_d = nullptr, *sums = nullptr;
    void *ws = nullptr;  // edge proposals of one batch: two u64 slots per candidate lane (or per linked result)
    const int saved_mode = idx->visited_mode;
    auto cleanup = [&]() {  // every exit below goes through here: workspaces freed, search state restored
        (void)hipFree(res_ids); (void)hipFree(res_d); (void)hipFree(indeg); (void)hipFree(sums); (void)hipFree(scratch);
        (void)hipFree(base_deg);
        roar_bk_ws_destroy(ws);
        idx->visited_mode = saved_mode;
        idx->building = false;
        idx->ip = idx->metric != ROAR_METRIC_L2;
    };
#define HIP_TRYC(expr)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            cleanup();                                                                              \
            return fail(ROAR_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_));                   \
        }                                                                                           \
    } while (0)
    HIP_TRYC(roar_dev_malloc((void **)&res_ids, (size_t)cap * K * 4));
    HIP_TRYC(roar_dev_malloc((void **)&res_d, (size_t)cap * K * 4));
    HIP_TRYC(roar_dev_malloc((void **)&indeg, (size_t)n * 4));
    HIP_TRYC(hipMemset(indeg, 0, (size_t)n * 4));
    HIP_TRYC(roar_dev_malloc((void **)&base_deg, (size_t)n * 4));
    HIP_TRYC(roar_dev_malloc((void **)&sums, idx->dim_pad * sizeof(float)));
    HIP_TRYC(roar_dev_malloc((void **)&scratch, 16));
    {
        const uint64_t per_list = 2ull * std::max<uint64_t>(std::max<uint64_t>(K, M), 1);
        const int rc = roar_bk_ws_create(std::max<uint64_t>(cap * per_list, n), &ws);
        if (rc != 0) {
            cleanup();
            return fail(ROAR_ERR_DEVICE, "out of device memory for the builder's edge proposals: %s", hipGetErrorString((hipError_t)rc));
        }
    }
#define BK_TRY(call)                                                                        \
    do {                                                                                    \
        int rc_ = (call);                                                                   \
        if (rc_ != 0) {                                                                     \
            cleanup();                                                                      \
            return fail(ROAR_ERR_DEVICE, "%s: %s", #call, hipGetErrorString((hipError_t)rc_)); \
        }                                                                                   \
    } while (0)
    // entry point: node nearest the centroid (CalculateProjectionep, src/index_bipartite.cpp:2004-2041)
    BK_TRY(roar_bk_entry_point(idx->base, idx->ld, idx->dim_pad, n, sums, (unsigned long long *)scratch, nullptr));
    unsigned long long best = 0;
    HIP_TRYC(hipMemcpy(&best, scratch, 8, hipMemcpyDeviceToHost));
    idx->ep = (uint32_t)(best & 0xFFFFFFFFu);
    const uint32_t seed = (uint32_t)std::min<uint64_t>(n, 1024);
    if (!bp->no_skeleton) BK_TRY(roar_bk_seed(idx->adj, S, seed, idx->ep, std::min<uint32_t>(M, seed - 1), nullptr));
    idx->visited_mode = ROAR_VISITED_LDS;
    idx->building = true;
    const bool saved_ip = idx->ip;
    auto search = [&](const float *q, uint64_t cnt, uint32_t qs, uint32_t k, uint32_t L) -> int {
        return roar_search_batch_device(idx, q, cnt, qs, k, L, res_ids, res_d, nullptr, nullptr, nullptr, nullptr, nullptr);
    };
    // phase 1 may link base points by L2 even when the index metric is IP: base-to-base inner
    // products send every node to the few large-norm points and leave the rest without in-edges.
    if (bp->phase1_l2) idx->ip = false;
    // phase 1: base-to-base links, rows capped at base_cap (skipped in the reference-shaped mode, which has
    // no skeleton: connectivity then comes from the supply pass alone, as in LinkProjection)
    const bool skeleton = !bp->no_skeleton;
    for (uint64_t a = skeleton ? seed : n; a < n;) {
        const uint64_t cnt = std::min<uint64_t>(std::min<uint64_t>(max_batch, a), n - a);
        if (C1) {  // insert with occlusion pruning of the node's own search results
            if ((e = search(idx->base + a * idx->ld, cnt, idx->ld, C1, std::max(L_build, C1)))) { cleanup(); return e; }
            BK_TRY(roar_bk_prune_link(ws, idx->adj, S, base_cap, idx->base, idx->ld, idx->dim_pad, idx->ip ? 1 : 0,
                                      (uint32_t)cnt, res_ids, C1, C1, M, indeg, (long long)a, 0, 1, idx->ep, 0, nullptr));
        } else {
            if ((e = search(idx->base + a * idx->ld, cnt, idx->ld, M, L_build))) { cleanup(); return e; }
            BK_TRY(roar_bk_link(ws, idx->adj, S, base_cap, (uint32_t)a, (uint32_t)cnt, res_ids, M, idx->ep, indeg, 0, nullptr));
        }
        a += cnt;
    }
    // seeds started with random links: give them real neighbours now that the graph exists
    if (!skeleton) {
    } else if (C1) {
        if ((e = search(idx->base, seed, idx->ld, C1, std::max(L_build, C1)))) { cleanup(); return e; }
        BK_TRY(roar_bk_prune_link(ws, idx->adj, S, base_cap, idx->base, idx->ld, idx->dim_pad, idx->ip ? 1 : 0, seed, res_ids,
                                  C1, C1, M, indeg, 0ll, 1, 1, 0xFFFFFFFFu, 0, nullptr));
    } else {
        if ((e = search(idx->base, seed, idx->ld, M, L_build))) { cleanup(); return e; }
        BK_TRY(roar_bk_link(ws, idx->adj, S, base_cap, 0, seed, res_ids, M, idx->ep, indeg, 1, nullptr));
    }
    idx->ip = saved_ip;
    BK_TRY(roar_bk_clamp(idx->adj, S, base_cap, n, nullptr));
    BK_TRY(roar_bk_snapshot_degree(idx->adj, S, max_degree, n, base_deg, nullptr));
    // projection pruning: 0 index metric (the reference's rule), 1 L2 geometry, 2 the list's own order, no occlusion
    const int proj_ip = (bp->proj_mode == 1) ? 0 : (idx->ip ? 1 : 0);
    const int proj_flags = (bp->proj_mode == 2) ? 3 : 0;
    const uint32_t proj_cap = bp->no_skeleton ? std::min(max_degree, Mp) : max_degree;
    // row_keep: max_degree is only the room in which a row collects proposals; after every phase rows longer than
    // row_keep keep their row_keep nearest neighbours (index metric)
    const uint32_t row_keep = (bp->row_keep && bp->row_keep < max_degree) ? bp->row_keep : 0;
    auto cut_rows = [&]() -> int {
        if (!row_keep) return 0;
        return roar_bk_cut_rows(idx->adj, S, max_degree, row_keep, idx->base, idx->ld, idx->dim_pad, saved_ip ? 1 : 0, n,
                                base_deg, nullptr);
    };
    // phase 2: projection of the training queries, `rounds` times over the improving graph
    // exact learn -> base kNN lists (the reference builder's input, LoadLearnBaseKNN,
    // src/index_bipartite.cpp:2622-2639) are projected first when the caller has them
    if (bp->learn_knn_dev && bp->n_knn) {
        for (uint64_t a = 0; a < bp->n_knn; a += max_batch) {
            const uint64_t cnt = std::min<uint64_t>(max_batch, bp->n_knn - a);
            BK_TRY(roar_bk_prune_link(ws, idx->adj, S, proj_cap, idx->base, idx->ld, idx->dim_pad, proj_ip, (uint32_t)cnt,
                                      bp->learn_knn_dev + a * C, C, C, Mp, indeg, -1ll, 0, 1, 0xFFFFFFFFu, proj_flags, nullptr));
            BK_TRY(roar_bk_clamp(idx->adj, S, max_degree, n, nullptr));
        }
        BK_TRY(cut_rows());
    }
    for (uint32_t r = 0; r < rounds && n_learn; ++r) {
        for (uint64_t a = 0; a < n_learn; a += max_batch) {
            const uint64_t cnt = std::min<uint64_t>(max_batch, n_learn - a);
            const uint32_t Lr = (r + 1 == rounds && bp->L_learn_last > L_learn) ? bp->L_learn_last : L_learn;
            if ((e = search(bp->learn_dev + a * bp->learn_stride_floats, cnt, bp->learn_stride_floats, C, Lr))) {
                cleanup();
                return e;
            }
            BK_TRY(roar_bk_prune_link(ws, idx->adj, S, max_degree, idx->base, idx->ld, idx->dim_pad, proj_ip, (uint32_t)cnt,
                                      res_ids, C, C, Mp, indeg, -1ll, 0, 1, 0xFFFFFFFFu, proj_flags, nullptr));
            BK_TRY(roar_bk_clamp(idx->adj, S, max_degree, n, nullptr));
        }
        BK_TRY(cut_rows());
    }
    // phase 3: connectivity supplement in the index metric
    if (bp->supply_k) {
        const uint32_t Ks = std::min<uint32_t>(bp->supply_k, K);
        const uint32_t Ls = std::max<uint32_t>(bp->supply_L, Ks);
        for (uint64_t a = 0; a < n;) {
            // without a skeleton the first nodes are linked one small batch at a time, each batch seeing
            // the supply edges of the ones before it (the reference walks the nodes one by one)
            const uint64_t cnt = std::min<uint64_t>(bp->no_skeleton ? std::min<uint64_t>(max_batch, std::max<uint64_t>(a / 2, 64))
                                                                   : max_batch, n - a);
            if (C3) {  // pruned supply edges: candidates = the node's search results not yet in its row
                if ((e = search(idx->base + a * idx->ld, cnt, idx->ld, C3, std::max(Ls, C3)))) { cleanup(); return e; }
                BK_TRY(roar_bk_prune_link(ws, idx->adj, S, max_degree, idx->base, idx->ld, idx->dim_pad,
                                          (bp->supply_mode == 1) ? 0 : (idx->ip ? 1 : 0), (uint32_t)cnt, res_ids, C3, C3, Ks, indeg,
                                          (long long)a, 1, 0, 0xFFFFFFFFu, (bp->supply_mode == 2) ? 3 : 0, nullptr));
            } else {
                if ((e = search(idx->base + a * idx->ld, cnt, idx->ld, Ks + 1, Ls))) { cleanup(); return e; }
                BK_TRY(roar_bk_supply(ws, idx->adj, S, max_degree, (uint32_t)a, (uint32_t)cnt, res_ids, Ks + 1, Ks, indeg, nullptr));
            }
            a += cnt;
            BK_TRY(roar_bk_clamp(idx->adj, S, max_degree, n, nullptr));
        }
    }
    BK_TRY(roar_bk_clamp(idx->adj, S, max_degree, n, nullptr));
    BK_TRY(cut_rows());
    // optional: drop phase-1 (skeleton) edges from rows that now hold enough projection / supply edges
    // (only when the skeleton was linked under another metric than the index's: an L2 index's skeleton IS
    // its graph -- trimming it on the SIFT shape drops recall from 0.95 to 0.42 at L_pq = 100)
    if (bp->trim_min_proj && bp->phase1_l2 && saved_ip) BK_TRY(roar_bk_trim(idx->adj, S, max_degree, base_deg, n, bp->trim_min_proj, bp->trim_keep_base, nullptr));
    for (uint32_t r = 0; r < 8; ++r) {
        BK_TRY(roar_bk_rescue(ws, idx->adj, S, max_degree, n, idx->ep, r, indeg, scratch, nullptr));
        uint32_t orphans = 0;
        HIP_TRYC(hipMemcpy(&orphans, scratch, 4, hipMemcpyDeviceToHost));
        if (orphans == 0) break;
    }
    BK_TRY(roar_bk_dedupe(idx->adj, S, max_degree, n, nullptr));
    BK_TRY(roar_bk_degree_stats(idx->adj, S, n, (unsigned long long *)scratch, scratch + 2, nullptr));
    unsigned long long st[2] = {0, 0};
    HIP_TRYC(hipMemcpy(st, scratch, 16, hipMemcpyDeviceToHost));
    idx->nedges = st[0];
    idx->max_deg = (uint32_t)(st[1] & 0xFFFFFFFFu);
    // rows collected proposals in a room of max_degree words; what is left needs round_up(longest row + 1, 32) words
    // per row (T2I-10M, room 384 / keep 128: 16.6 GB -> 6.4 GB)
    {
        const uint32_t S_new = round_up(idx->max_deg + 1, 32);
        if (S_new < S) {
            uint32_t *compact = nullptr;
            HIP_TRYC(roar_dev_malloc((void **)&compact, ((size_t)n * S_new + ADJ_SLACK_WORDS) * sizeof(uint32_t)));
            BK_TRY(roar_bk_restride(idx->adj, S, compact, S_new, n, nullptr));
            HIP_TRYC(hipMemsetAsync(compact + (size_t)n * S_new, 0, ADJ_SLACK_WORDS * sizeof(uint32_t), nullptr));
            HIP_TRYC(hipDeviceSynchronize());
            (void)hipFree(idx->adj);
            idx->adj = compact;
            idx->adj_stride = S_new;
        }
    }
#undef BK_TRY
#undef HIP_TRYC
    cleanup();
    return ROAR_OK;
}

// BuildRoarGraph in its batch-synchronous device form (kernels and the mapping to the reference's steps: roar_rgbuild.hip)
int roar_build_roargraph_device(roar_index *idx, const uint32_t *lists_dev, uint64_t n_learn, uint32_t K, uint32_t M_sq,
                                uint32_t M_pjbp, uint32_t L_pjpq) {
    DeviceRestore keep_device_;
    if (int busy_ = refuse_while_streaming(idx)) return busy_;
    if (!idx || !lists_dev || n_learn == 0 || K < 2) return fail(ROAR_ERR_ARG, "learn base knn file error");
    if (M_pjbp == 0 || M_pjbp > 128) return fail(ROAR_ERR_ARG, "M_pjbp must be in 1..128");
    if (!idx->base || idx->n < 2) return fail(ROAR_ERR_ARG, "BuildRoarGraph needs the base vectors first");
    if (idx->n >= (1ull << 31) || n_learn >= (1ull << 32) - 1) return fail(ROAR_ERR_ARG, "too many points or training queries");
    const uint32_t C = std::min<uint32_t>(std::min<uint32_t>(M_sq ? M_sq : K, K), 255u);  // candidates per training query
    if (C < 2) return fail(ROAR_ERR_ARG, "M_sq must be at least 2");
    const uint32_t M = M_pjbp;
    const uint32_t Lp = std::max<uint32_t>(L_pjpq, M);
    if (8ull * (Lp + 1) > 120 * 1024) return fail(ROAR_ERR_ARG, "L_pjpq = %u does not fit the search kernel's LDS pool", L_pjpq);
    int e = use_device(idx);
    if (e) return e;
    std::lock_guard<std::mutex> guard(idx->mu);
    free_graph(idx);
    const uint64_t n = idx->n;
    const int ip = idx->ip ? 1 : 0;
    const uint32_t Sg = round_up(M + 1, 32), Ss = round_up(2 * M + 1, 32), Sf = round_up(3 * M + 1, 32);
    const uint64_t batch_lists = std::min<uint64_t>(n_learn, 1u << 20);      // projection: training queries per batch
    const uint64_t batch_nodes = std::min<uint64_t>(n, 1u << 16);            // connectivity: nodes per batch (k = L_pjpq results each)
    const uint64_t max_props = std::max<uint64_t>(std::max<uint64_t>(batch_lists, batch_nodes) * M, std::min<uint64_t>(n, 1u << 20) * M);
    const uint32_t scratch_waves = 4096;

    uint32_t *grows = nullptr, *srows = nullptr, *final_adj = nullptr, *last = nullptr, *out = nullptr, *cnt = nullptr;
    uint32_t *heads = nullptr, *heads_sorted = nullptr, *nheads_dev = nullptr, *res_ids = nullptr, *scratch = nullptr;
    float *res_d = nullptr, *sums = nullptr;
    unsigned long long *props_a = nullptr, *props_b = nullptr, *keyscratch = nullptr;
    void *sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    const int saved_mode = idx->visited_mode;
    auto cleanup = [&]() {
        (void)hipFree(grows); (void)hipFree(srows); (void)hipFree(final_adj); (void)hipFree(last); (void)hipFree(out);
        (void)hipFree(cnt); (void)hipFree(heads); (void)hipFree(heads_sorted); (void)hipFree(nheads_dev); (void)hipFree(res_ids);
        (void)hipFree(res_d); (void)hipFree(sums); (void)hipFree(scratch); (void)hipFree(props_a); (void)hipFree(props_b);
        (void)hipFree(keyscratch); (void)hipFree(sort_tmp);
        idx->visited_mode = saved_mode;
        idx->building = false;
    };
    auto drop_graph = [&]() {  // a failed build leaves no half-made graph behind (idx->adj may alias one of the buffers above)
        idx->adj = nullptr;
        idx->gn = idx->nedges = 0;
    };
#define RG_HIP(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            drop_graph();                                                                   \
            cleanup();                                                                      \
            return fail(ROAR_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_));           \
        }                                                                                   \
    } while (0)
#define RG_TRY(call)                                                                        \
    do {                                                                                    \
        int rc_ = (call);                                                                   \
        if (rc_ != 0) {                                                                     \
            drop_graph();                                                                   \
            cleanup();                                                                      \
            return fail(ROAR_ERR_DEVICE, "%s: %s", #call, hipGetErrorString((hipError_t)rc_)); \
        }                                                                                   \
    } while (0)
    RG_HIP(roar_dev_malloc((void **)&grows, ((size_t)n * Sg + ADJ_SLACK_WORDS) * 4));
    RG_HIP(hipMemset(grows, 0, ((size_t)n * Sg + ADJ_SLACK_WORDS) * 4));
    RG_HIP(roar_dev_malloc((void **)&srows, ((size_t)n * Ss + ADJ_SLACK_WORDS) * 4));
    RG_HIP(hipMemset(srows, 0, ((size_t)n * Ss + ADJ_SLACK_WORDS) * 4));
    RG_HIP(roar_dev_malloc((void **)&last, (size_t)n * 4));
    RG_HIP(hipMemset(last, 0, (size_t)n * 4));
    const uint64_t max_lists = std::max<uint64_t>(std::max<uint64_t>(batch_lists, batch_nodes), std::min<uint64_t>(n, 1u << 20));
    RG_HIP(roar_dev_malloc((void **)&out, (size_t)max_lists * M * 4));
    RG_HIP(roar_dev_malloc((void **)&cnt, (size_t)max_lists * 4));
    RG_HIP(roar_dev_malloc((void **)&props_a, (size_t)max_props * 8));
    RG_HIP(roar_dev_malloc((void **)&props_b, (size_t)max_props * 8));
    RG_HIP(roar_dev_malloc((void **)&heads, (size_t)std::min<uint64_t>(max_props, n) * 4 + 64));
    RG_HIP(roar_dev_malloc((void **)&heads_sorted, (size_t)std::min<uint64_t>(max_props, n) * 4 + 64));
    RG_HIP(roar_dev_malloc((void **)&nheads_dev, 4));
    RG_TRY(roar_rg_sort_tmp_bytes(max_props, &sort_tmp_bytes));
    RG_HIP(roar_dev_malloc(&sort_tmp, sort_tmp_bytes ? sort_tmp_bytes : 16));
    RG_HIP(roar_dev_malloc((void **)&res_ids, (size_t)batch_nodes * Lp * 4));
    RG_HIP(roar_dev_malloc((void **)&res_d, (size_t)batch_nodes * Lp * 4));
    RG_HIP(roar_dev_malloc((void **)&keyscratch, (size_t)scratch_waves * Lp * 8));
    RG_HIP(roar_dev_malloc((void **)&sums, idx->dim_pad * sizeof(float)));
    RG_HIP(roar_dev_malloc((void **)&scratch, 16));

    // the lists come from the caller's device memory: every id the projection reads must name a base point
    {
        uint32_t bad = 0;
        RG_HIP(hipMemset(scratch, 0, 16));
        for (uint64_t a = 0; a < n_learn; a += batch_lists)
            RG_TRY(roar_rg_check_lists(lists_dev + a * K, K, C, std::min<uint64_t>(batch_lists, n_learn - a), (uint32_t)n, scratch, nullptr));
        RG_HIP(hipMemcpy(&bad, scratch, 4, hipMemcpyDeviceToHost));
        if (bad) {
            drop_graph();
            cleanup();
            return fail(ROAR_ERR_FORMAT, "learn base knn file error");  // (an id outside the base set)
        }
    }
    // CalculateProjectionep (:2004-2041)
    RG_TRY(roar_bk_entry_point(idx->base, idx->ld, idx->dim_pad, n, sums, (unsigned long long *)scratch, nullptr));
    unsigned long long best = 0;
    RG_HIP(hipMemcpy(&best, scratch, 8, hipMemcpyDeviceToHost));
    idx->ep = (uint32_t)(best & 0xFFFFFFFFu);

    auto reverse = [&](uint32_t *rows, uint32_t S, uint32_t cap, int list_backfill, uint64_t nprops) -> int {
        return roar_rg_reverse(idx->base, idx->ld, idx->dim_pad, ip, rows, S, cap, M, list_backfill, props_a, props_b, nprops,
                               sort_tmp, sort_tmp_bytes, heads, heads_sorted, nheads_dev, nullptr);
    };
    // ---- projection (:1059-1129)
    for (uint64_t a = 0; a < n_learn; a += batch_lists)
        RG_TRY(roar_rg_last_list(lists_dev + a * K, K, (uint32_t)std::min<uint64_t>(batch_lists, n_learn - a), (uint32_t)a, last, nullptr));
    for (uint64_t a = 0; a < n_learn; a += batch_lists) {
        const uint32_t c = (uint32_t)std::min<uint64_t>(batch_lists, n_learn - a);
        RG_TRY(roar_rg_list_prune(idx->base, idx->ld, idx->dim_pad, ip, 0, lists_dev + a * K, K, C, c, 0, nullptr, 0, nullptr, 0, M,
                                  0, out, cnt, nullptr));
        RG_TRY(roar_rg_set_rows(grows, Sg, lists_dev + a * K, K, c, (uint32_t)a, 0, 1, last, out, cnt, M, props_a, nullptr));
        RG_TRY(reverse(grows, Sg, M, 1, (uint64_t)c * M));
    }
    for (uint64_t v0 = 0; v0 < n; v0 += (1u << 20)) {  // :1096-1099: every node's row proposes the node once more
        const uint64_t c = std::min<uint64_t>(1u << 20, n - v0);
        RG_TRY(roar_rg_props_from_rows(grows, Sg, v0, c, M, props_a, nullptr));
        RG_TRY(reverse(grows, Sg, M, 1, c * M));
    }
    // ---- connectivity enhancement (:1183-1262): the supply graph starts as the projection graph and is what gets searched
    RG_TRY(roar_rg_copy_rows(grows, Sg, srows, Ss, n, nullptr));
    idx->adj = srows;
    idx->adj_stride = Ss;
    idx->gn = n;
    idx->max_deg = 2 * M;
    idx->visited_mode = ROAR_VISITED_LDS;
    idx->building = true;
    for (uint64_t a = 0; a < n;) {
        // the reference walks the nodes one by one, each seeing the supply edges of the ones before it: small batches first
        const uint64_t c = std::min<uint64_t>(std::min<uint64_t>(batch_nodes, std::max<uint64_t>(a / 2, 64)), n - a);
        int rc = roar_search_batch_device(idx, idx->base + a * idx->ld, c, idx->ld, Lp, Lp, res_ids, res_d, nullptr, nullptr,
                                          nullptr, nullptr, nullptr);
        if (rc) {
            const std::string keep = g_err;
            drop_graph();
            cleanup();
            g_err = keep;
            return rc;
        }
        RG_TRY(roar_rg_supply_prune(idx->base, idx->ld, idx->dim_pad, ip, res_ids, res_d, Lp, (uint32_t)c, (uint32_t)a, grows, Sg,
                                    M, keyscratch, scratch_waves, out, cnt, nullptr));
        RG_TRY(roar_rg_set_rows(srows, Ss, nullptr, 0, (uint32_t)c, 0, (uint32_t)a, 0, nullptr, out, cnt, M, props_a, nullptr));
        RG_TRY(reverse(srows, Ss, 2 * M, 0, c * M));
        a += c;
    }
    idx->building = false;
    idx->visited_mode = saved_mode;
    for (uint64_t v0 = 0; v0 < n; v0 += (1u << 20)) {  // :1230-1258: supply rows longer than M are pruned once more
        const uint32_t c = (uint32_t)std::min<uint64_t>(1u << 20, n - v0);
        RG_TRY(roar_rg_list_prune(idx->base, idx->ld, idx->dim_pad, ip, 1, nullptr, 0, 0, c, (uint32_t)v0, srows, Ss, grows, Sg, M, M,
                                  out, cnt, nullptr));
        RG_TRY(roar_rg_set_rows(srows, Ss, nullptr, 0, c, 0, (uint32_t)v0, 0, nullptr, out, cnt, M, props_a, nullptr));
    }
    // ---- merge (:1262-1275)
    RG_HIP(roar_dev_malloc((void **)&final_adj, ((size_t)n * Sf + ADJ_SLACK_WORDS) * 4));
    RG_HIP(hipMemset(final_adj, 0, ((size_t)n * Sf + ADJ_SLACK_WORDS) * 4));
    RG_TRY(roar_rg_merge_rows(grows, Sg, srows, Ss, final_adj, Sf, n, M, nullptr));
    RG_TRY(roar_bk_degree_stats(final_adj, Sf, n, (unsigned long long *)scratch, scratch + 2, nullptr));
    unsigned long long st[2] = {0, 0};
    RG_HIP(hipMemcpy(st, scratch, 16, hipMemcpyDeviceToHost));
    idx->adj = final_adj;
    final_adj = nullptr;  // owned by the index now
    idx->adj_stride = Sf;
    idx->nedges = st[0];
    idx->max_deg = (uint32_t)(st[1] & 0xFFFFFFFFu);
    idx->graph_hash_valid = false;
#undef RG_HIP
#undef RG_TRY
    cleanup();
    return ROAR_OK;
}

int roar_build_roargraph(roar_index *idx, const uint32_t *learn_knn, uint64_t n_learn, uint32_t K, uint32_t M_sq,
                         uint32_t M_pjbp, uint32_t L_pjpq) {
    DeviceRestore keep_device_;
    if (int busy_ = refuse_while_streaming(idx)) return busy_;
    if (!idx || !learn_knn || n_learn == 0 || K < 2) return fail(ROAR_ERR_ARG, "learn base knn file error");
    if (!idx->base || idx->n < 2) return fail(ROAR_ERR_ARG, "BuildRoarGraph needs the base vectors first");
    const uint32_t C = std::min<uint32_t>(std::min<uint32_t>(M_sq ? M_sq : K, K), 255u);
    for (uint64_t i = 0; i < n_learn; ++i)
        for (uint32_t j = 0; j < C; ++j)
            if (learn_knn[i * K + j] >= idx->n) return fail(ROAR_ERR_FORMAT, "learn base knn file error");
    int e = use_device(idx);
    if (e) return e;
    uint32_t *lists_dev = nullptr;
    HIP_TRY(roar_dev_malloc((void **)&lists_dev, (size_t)n_learn * K * sizeof(uint32_t)));
    hipError_t he = hipMemcpy(lists_dev, learn_knn, (size_t)n_learn * K * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (he != hipSuccess) {
        (void)hipFree(lists_dev);
        return fail(ROAR_ERR_DEVICE, "upload of the kNN lists failed: %s", hipGetErrorString(he));
    }
    e = roar_build_roargraph_device(idx, lists_dev, n_learn, K, M_sq, M_pjbp, L_pjpq);
    (void)hipFree(lists_dev);
    return e;
}
}  // extern "C"
