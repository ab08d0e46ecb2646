// roar_capi.cpp -- C ABI (include/roargraph_amd.h) over the HIP kernels: index object, loaders
// for the reference's on-disk formats, HBM layout, launch configuration.  Host-only code built
// with hipcc.  There is deliberately no CPU compute path in this library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/roargraph_amd.h"
#include "roar_device.h"

extern "C" {  // roar_build.hip
int roar_bk_entry_point(const float *base, uint64_t ld, uint32_t dim, uint64_t n, float *sums_dev,
                        unsigned long long *best_dev, void *stream);
int roar_bk_seed(uint32_t *adj, uint32_t S, uint32_t seed, uint32_t ep, uint32_t M, void *stream);
int roar_bk_ws_create(uint64_t max_props, void **out);
void roar_bk_ws_destroy(void *ws);
int roar_bk_link(void *ws, uint32_t *adj, uint32_t S, uint32_t max_deg, uint32_t a, uint32_t count, const uint32_t *res,
                 uint32_t M, uint32_t ep, uint32_t *indeg, int relink, void *stream);
int roar_bk_prune_link(void *ws, uint32_t *adj, uint32_t S, uint32_t cap, const float *base, uint64_t ld, uint32_t dim,
                       int metric_ip, uint32_t count, const uint32_t *res, uint32_t M, uint32_t C, uint32_t Mp,
                       uint32_t *indeg, long long target_base, int skip_existing, int backfill, uint32_t skip_node, int flags,
                       void *stream);
int roar_bk_supply(void *ws, uint32_t *adj, uint32_t S, uint32_t max_deg, uint32_t a, uint32_t count, const uint32_t *res,
                   uint32_t M, uint32_t Kc, uint32_t *indeg, void *stream);
int roar_bk_snapshot_degree(const uint32_t *adj, uint32_t S, uint32_t max_deg, uint64_t n, uint32_t *base_deg,
                            void *stream);
int roar_bk_clamp(uint32_t *adj, uint32_t S, uint32_t max_deg, uint64_t n, void *stream);
int roar_bk_rescue(void *ws, uint32_t *adj, uint32_t S, uint32_t max_deg, uint64_t n, uint32_t ep, uint32_t r,
                   uint32_t *indeg, uint32_t *orphans, void *stream);
int roar_bk_trim(uint32_t *adj, uint32_t S, uint32_t max_deg, const uint32_t *base_deg, uint64_t n, uint32_t min_proj,
                 uint32_t keep_base, void *stream);
int roar_bk_dedupe(uint32_t *adj, uint32_t S, uint32_t max_deg, uint64_t n, void *stream);
int roar_bk_cut_rows(uint32_t *adj, uint32_t S, uint32_t cap, uint32_t keep, const float *base, uint64_t ld, uint32_t dim,
                     int metric_ip, uint64_t n, uint32_t *base_deg, void *stream);
int roar_bk_degree_stats(const uint32_t *adj, uint32_t S, uint64_t n, unsigned long long *sum_dev, uint32_t *max_dev,
                         void *stream);
int roar_bk_graph_hash(const uint32_t *adj, uint32_t S, uint64_t n, uint32_t ep, unsigned long long *out_dev, void *stream);
int roar_bk_restride(const uint32_t *src, uint32_t S_old, uint32_t *dst, uint32_t S_new, uint64_t n, void *stream);
}

extern "C" {  // roar_rgbuild.hip
int roar_rg_list_prune(const float *base, uint64_t ld, uint32_t dim, int ip, int mode, const uint32_t *lists, uint32_t K,
                       uint32_t C, uint32_t count, uint32_t t0, const uint32_t *rows, uint32_t S, const uint32_t *grows,
                       uint32_t Sg, uint32_t M, uint32_t only_longer_than, uint32_t *out, uint32_t *cnt, void *stream);
int roar_rg_supply_prune(const float *base, uint64_t ld, uint32_t dim, int ip, const uint32_t *res_ids, const float *res_d,
                         uint32_t Lres, uint32_t count, uint32_t t0, const uint32_t *grows, uint32_t Sg, uint32_t M,
                         unsigned long long *keyscratch, uint32_t scratch_waves, uint32_t *out, uint32_t *cnt, void *stream);
int roar_rg_check_lists(const uint32_t *lists, uint32_t K, uint32_t C, uint64_t count, uint32_t n, uint32_t *bad_dev, void *stream);
int roar_rg_last_list(const uint32_t *lists, uint32_t K, uint32_t count, uint32_t sq0, uint32_t *last, void *stream);
int roar_rg_set_rows(uint32_t *rows, uint32_t S, const uint32_t *lists, uint32_t K, uint32_t count, uint32_t sq0, uint32_t t0,
                     int by_pivot, const uint32_t *last, const uint32_t *out, const uint32_t *cnt, uint32_t M,
                     unsigned long long *props, void *stream);
int roar_rg_props_from_rows(const uint32_t *rows, uint32_t S, uint64_t v0, uint64_t count, uint32_t M, unsigned long long *props,
                            void *stream);
int roar_rg_reverse(const float *base, uint64_t ld, uint32_t dim, int ip, uint32_t *rows, uint32_t S, uint32_t cap, uint32_t M,
                    int list_backfill, unsigned long long *keys_in, unsigned long long *keys_out, uint64_t nprops, void *sort_tmp,
                    size_t sort_tmp_bytes, uint32_t *heads, uint32_t *heads_sorted, uint32_t *nheads_dev, void *stream);
int roar_rg_sort_tmp_bytes(uint64_t max_keys, size_t *bytes);
int roar_rg_merge_rows(const uint32_t *grows, uint32_t Sg, const uint32_t *srows, uint32_t Ss, uint32_t *adj, uint32_t Sf,
                       uint64_t n, uint32_t M, void *stream);
int roar_rg_copy_rows(const uint32_t *src, uint32_t Ssrc, uint32_t *dst, uint32_t Sdst, uint64_t n, void *stream);
}
extern "C" int roar_hub_build_rows(const uint32_t *adj, uint32_t S, uint64_t n, uint32_t *out, uint32_t Sh, uint32_t *indeg,
                                   uint16_t *slot_of, unsigned long long *keys_a, unsigned long long *keys_b, void *sort_tmp,
                                   size_t *sort_tmp_bytes, void *stream);  // roar_hub.hip
extern "C" int roar_launch_queue_trace(uint32_t capacity, uint32_t n_ops, const uint8_t *ops, const uint32_t *ids,
                                       const float *dists, int form, uint32_t *snap, void *stream);
extern "C" int roar_knn_run(const float *q, uint64_t ldq, uint32_t nq, const float *b, uint64_t ldb, uint64_t n, uint32_t dim,
                            int ip, uint32_t k, uint32_t cap, void *cand, void *cursor, void *tau, void *bnorm,
                            void *overflow_dev, uint32_t *ids, float *dists, uint32_t *overflowed, void *stream);
extern "C" {  // roar_knn_bf16.hip
uint32_t roar_knn_bf16_row_quads(uint32_t dim);
int roar_knn_bf16_convert(const float *src, uint64_t ld, uint32_t dim, uint64_t n, uint64_t n_pad, void *dst, float *sqnorm, uint32_t *max_sq_bits,
                          void *stream);
int roar_knn_run_bf16(const float *q, uint64_t ldq, uint32_t nq, const float *b, uint64_t ldb, uint64_t n, uint32_t dim, int ip, uint32_t k,
                      uint32_t cap, const void *bb, const float *bnorm, void *qb, void *cand, void *cursor, void *tau, void *ea, void *kq,
                      void *overflow_dev, uint32_t *rerank_rows_dev, uint32_t *ids, float *dists, uint32_t *overflowed, void *stream);
}

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(ROAR_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

inline uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

// Every device allocation of this file goes through here so that error paths can be exercised: the n-th allocation from
// now fails with hipErrorOutOfMemory when armed (roar_debug_fail_device_alloc_at, or ROAR_FAIL_AT=n in the environment
// when the library is loaded).  Unarmed it is hipMalloc.
std::atomic<int64_t> g_fail_alloc_in{[] {
    const char *v = getenv("ROAR_FAIL_AT");
    return (int64_t)(v ? atoll(v) : 0);
}()};
hipError_t roar_dev_malloc(void **p, size_t bytes) {
    if (g_fail_alloc_in.load(std::memory_order_relaxed) > 0 && g_fail_alloc_in.fetch_sub(1) == 1) {
        *p = nullptr;
        return hipErrorOutOfMemory;
    }
    return hipMalloc(p, bytes);
}
template <typename T>
hipError_t roar_dev_malloc(T **p, size_t bytes) {
    return roar_dev_malloc(reinterpret_cast<void **>(p), bytes);
}

}  // namespace

struct roar_index {
    int device = 0;
    uint32_t dim = 0, dim_pad = 0;
    int metric = ROAR_METRIC_L2;
    bool ip = false, need_normalize = false;
    int n_cus = 0;
    // base rows in HBM
    float *base = nullptr;
    bool base_owned = false;
    // 8-bit screening copy of the base rows (RoarSearchParams::screen), n x roar_screen8_stride(dim) bytes, made on first use
    unsigned char *screen = nullptr;
    float *screen_mu = nullptr;  // inner product: the mean of the base rows, which the copy's codes are centred on (dim_pad floats)
    uint64_t screen_n = 0;
    int screen_mode = 0;  // option "screen": 0 auto (on where it pays), 1 off, 2 on wherever the kernels have it
    int launch_events = 1;  // option "launch_events": 0 = no timing events around a launch ("last_kernel_ns" then keeps its last value)
    int last_screen = 0;
    // What a launch plan leaves open when no option fixes it -- screen on / off, visited form, table size -- is MEASURED on the loaded
    // index: the first search of a (L_pq band, batch-size band, grid cap) times the candidates on the first queries of that call and
    // the winner is kept here (calibrate_plan).  Option "calibrate" = 0: fixed defaults instead.
    struct Calibration {
        uint64_t key;
        int screen, mode;
        uint32_t entries;
        int protect;
        float best_ms, default_ms;
        uint32_t sample, L, runs;
    };
    std::vector<Calibration> plan_cache;
    int calibrate = 1;
    int last_calibrated = 0;
    int knn_mode = 0;  // option "knn_mode": 0 auto (bf16 MFMA screen + exact f32 re-rank where the row length allows), 1 f32 MFMA only
    int last_knn_bf16 = 0;
    int64_t last_knn_rerank_rows = 0;  // rows that reached the exact re-rank in the last roar_exact_knn_device call, all queries together
    uint64_t n = 0;
    uint32_t ld = 0;
    // graph in HBM (fixed-stride rows) + host CSR copy for save / get
    uint32_t *adj = nullptr;
    uint32_t adj_stride = 0, ep = 0, max_deg = 0;
    uint64_t gn = 0, nedges = 0;
    // hub-annotated copy of the rows for long pools (built on first use): the 65 535 highest-in-degree nodes get a slot
    // in an exact on-chip visited bitmap; rows list their hub neighbours first and carry the slots behind the ids
    uint32_t *adj_hub = nullptr;
    uint32_t hub_stride = 0;
    int hub_state = 0;   // 0 not built, 1 built, -1 not applicable to this graph (a row longer than 170)
    int hub_mode = 0;    // option: 0 auto, 1 never, 2 always (when applicable)
    uint64_t graph_hash = 0;  // of the device rows (degree + neighbours in order, per row) and the entry point
    bool graph_hash_valid = false;
    std::vector<uint64_t> h_off;
    std::vector<uint32_t> h_nbrs;
    // bipartite graph (base nodes [0, n), training-query nodes [n, bp_n)): CSR in HBM + host copy
    uint64_t *bp_off = nullptr;
    uint32_t *bp_nbrs = nullptr;
    uint64_t bp_n = 0, bp_edges = 0;
    bool bp_checked = false;
    std::vector<uint64_t> hb_off;
    std::vector<uint32_t> hb_nbrs;
    uint32_t *init_dev = nullptr;
    uint64_t init_cap = 0;
    uint32_t *tags32 = nullptr;   // eight-waves-per-query form: u32 epoch tags, one list per resident workgroup
    uint32_t *epochs32 = nullptr;
    uint32_t slots32 = 0;
    uint64_t tag_stride32 = 0;
    uint64_t bp_seed = 0x9E3779B97F4A7C15ull;
    // options
    int visited_mode = ROAR_VISITED_AUTO;
    int last_mode = ROAR_VISITED_LDS;
    int waves_per_cu = 0;
    int vis_entries = 0;
    int tag_protect = -1;  // option: the forgetful tag table of an LDS-pool launch keeps the tags of pool members (roar_wave.h TagProtect); -1 = measured
    int last_tag_protect = 0;
    int max_grid = 0;
    int search_kernel = 0;  // 0 auto, 1 one wave per query, 2 four waves per query, 3 two waves per query
    int pool_form = 0;      // 0 auto (register pool for L_pq <= 64), 1 always the LDS pool
    int profile = 0;        // 1: instrumented kernel, phase timers readable with roar_last_profile
    unsigned long long *prof_dev = nullptr;
    uint64_t prof_cap = 0, prof_nq = 0;
    // caller's per-query traffic counters (roar_set_traffic_counters_device): u32[traffic_cap][4], not owned
    uint32_t *traffic_dev = nullptr;
    uint64_t traffic_cap = 0;
    bool building = false;  // searches issued by roar_build_graph run under their own kernel symbol
    RoarKernelSel last_sel = {};
    // workspaces
    uint32_t *counters = nullptr;  // two-hop kernels: [0] dequeue counter (zeroed before every launch)
    // SearchRoarGraph launches: a ring of 16-byte launch cells (RoarSearchParams::counter), zero between launches -- the
    // kernel cleans up after itself, so launches on different streams may overlap and nothing is queued in front of one
    uint32_t *ring = nullptr;
    std::atomic<uint32_t> launch_seq{0};
    uint8_t *tags = nullptr;
    uint32_t *epochs = nullptr;
    uint32_t slots = 0;
    uint64_t tag_stride = 0;
    // LDS mode, exact form: spill lists, one u16[n] list per resident workgroup + its epoch (roar_search_v2.h visited_ldsn)
    uint16_t *spill = nullptr;
    uint32_t *spill_epochs = nullptr;
    uint32_t spill_slots = 0;
    uint64_t spill_stride = 0;  // elements
    int last_forgetful = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ev_valid = false;
    hipStream_t last_stream = nullptr;  // stream of the most recent search launch, and whether it used the HBM tag lists
    bool last_used_tags = false;
    // host-API staging: queries are packed into pinned host memory that the kernel reads in place (no H2D step),
    // results come back through a pinned block with one asynchronous copy; everything on the index's own stream
    hipStream_t stream = nullptr;
    float *q_pin = nullptr;      // pinned + mapped
    float *q_pin_dev = nullptr;  // its device address
    uint32_t *ready_pin = nullptr, *ready_dev = nullptr;  // rows of q_pin packed so far (read by a running launch)
    uint64_t q_cap = 0;
    uint32_t *out_dev = nullptr;
    uint32_t *out_pin = nullptr;
    uint64_t out_cap = 0;
    // SearchRoarGraph is re-entrant in the reference (called from num_threads OpenMP threads,
    // tests/test_search_roargraph.cpp:203); the host-buffer entry points share staging buffers, the
    // dequeue counter and the timing events, so concurrent callers are serialised here.
    std::mutex mu;
    // single-query calls from concurrent host threads are combined into one launch (see search_one)
    struct OneQuery {
        const float *q;
        uint32_t qstride, k, L;
        uint32_t *ids;
        float *dists;
        uint32_t *cmps, *hops, *scanned;
        int rc = 0;
        std::string err;
        bool done = false;
    };
    std::mutex cq_mu;
    std::condition_variable cq_cv;
    std::vector<OneQuery *> cq;
    bool cq_leader = false;
    int combine = 1;
    int64_t last_combined = 0;
    // last launch
    int64_t last_grid = 0, last_lds = 0, last_vis_entries = 0;
    std::vector<std::pair<uint64_t, int>> occ_cache;  // (kernel form, LDS bytes) -> resident workgroups per CU
};

namespace {

// Every entry point that selects a device puts the calling thread's current device back when it returns: a caller that
// also uses torch / HIP on another device must not find its context moved (a replica lives on its own device).
struct DeviceRestore {
    int prev = -1;
    DeviceRestore() {
        if (hipGetDevice(&prev) != hipSuccess) {
            (void)hipGetLastError();
            prev = -1;
        }
    }
    ~DeviceRestore() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceRestore(const DeviceRestore &) = delete;
    DeviceRestore &operator=(const DeviceRestore &) = delete;
};

int use_device(roar_index *idx) {
    HIP_TRY(hipSetDevice(idx->device));
    return ROAR_OK;
}

void free_graph(roar_index *idx) {
    idx->plan_cache.clear();  // (measured on the graph that goes away)
    if (idx->adj) (void)hipFree(idx->adj);
    idx->adj = nullptr;
    if (idx->adj_hub) (void)hipFree(idx->adj_hub);
    idx->adj_hub = nullptr;
    idx->hub_state = 0;
    idx->h_off.clear();
    idx->h_nbrs.clear();
    idx->gn = idx->nedges = 0;
    idx->graph_hash_valid = false;
}

void free_bipartite(roar_index *idx) {
    if (idx->bp_off) (void)hipFree(idx->bp_off);
    if (idx->bp_nbrs) (void)hipFree(idx->bp_nbrs);
    idx->bp_off = nullptr;
    idx->bp_nbrs = nullptr;
    idx->hb_off.clear();
    idx->hb_nbrs.clear();
    idx->bp_n = idx->bp_edges = 0;
    idx->bp_checked = false;
}

void free_base(roar_index *idx) {
    idx->plan_cache.clear();
    if (idx->screen) (void)hipFree(idx->screen);
    if (idx->screen_mu) (void)hipFree(idx->screen_mu);
    idx->screen = nullptr;
    idx->screen_mu = nullptr;
    idx->screen_n = 0;
    if (idx->base && idx->base_owned) (void)hipFree(idx->base);
    idx->base = nullptr;
    idx->base_owned = false;
    idx->n = 0;
    idx->bp_checked = false;  // the bipartite split point is the number of base rows
}

void free_tags(roar_index *idx) {
    if (idx->tags32) (void)hipFree(idx->tags32);
    if (idx->epochs32) (void)hipFree(idx->epochs32);
    idx->tags32 = nullptr;
    idx->epochs32 = nullptr;
    idx->slots32 = 0;
    if (idx->tags) (void)hipFree(idx->tags);
    if (idx->epochs) (void)hipFree(idx->epochs);
    idx->tags = nullptr;
    idx->epochs = nullptr;
    idx->slots = 0;
    if (idx->spill) (void)hipFree(idx->spill);
    if (idx->spill_epochs) (void)hipFree(idx->spill_epochs);
    idx->spill = nullptr;
    idx->spill_epochs = nullptr;
    idx->spill_slots = 0;
}

// util.h:214-225
void normalize_row(float *arr, size_t dim) {
    float sum = 0.0f;
    for (size_t i = 0; i < dim; ++i) sum += arr[i] * arr[i];
    sum = std::sqrt(sum);
    for (size_t i = 0; i < dim; ++i) arr[i] = arr[i] / sum;
}

int alloc_base(roar_index *idx, uint64_t n) {
    free_base(idx);
    free_tags(idx);
    idx->ld = idx->dim_pad;
    HIP_TRY(roar_dev_malloc((void **)&idx->base, (size_t)n * idx->ld * sizeof(float)));
    idx->base_owned = true;
    idx->n = n;
    return ROAR_OK;
}

// Upload host rows [r0, r0+cnt) already padded to dim_pad.
int upload_rows(roar_index *idx, const float *padded, uint64_t r0, uint64_t cnt) {
    HIP_TRY(hipMemcpy(idx->base + r0 * idx->ld, padded, (size_t)cnt * idx->ld * sizeof(float), hipMemcpyHostToDevice));
    return ROAR_OK;
}

// Visited-tag pools cost slots x n bytes (u8 tags) or slots x 4 n bytes (u32 tags) of HBM.  A pool never takes more
// than half of what is free: the launch then runs with fewer resident workgroups instead of failing (a shard of
// 100 M points still searches in EXACT mode, with a smaller grid).  *slots is lowered to what was granted.
uint32_t affordable_slots(uint64_t bytes_per_slot, uint64_t already_held, uint32_t want) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return want;
    const uint64_t budget = ((uint64_t)free_b + already_held) / 2;
    const uint64_t fit = budget / std::max<uint64_t>(bytes_per_slot, 1);
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(want, fit));
}

int ensure_tags(roar_index *idx, uint32_t *slots_io) {
    uint32_t slots = *slots_io;
    if (idx->tags && idx->slots >= slots) return ROAR_OK;
    const uint64_t stride = std::max<uint64_t>(16, ((idx->n + 15) / 16) * 16);
    slots = affordable_slots(stride, idx->tags ? (uint64_t)idx->slots * idx->tag_stride : 0, slots);
    *slots_io = slots;
    if (idx->tags && idx->slots >= slots) return ROAR_OK;
    free_tags(idx);
    idx->tag_stride = stride;
    HIP_TRY(roar_dev_malloc((void **)&idx->tags, (size_t)slots * idx->tag_stride));
    HIP_TRY(hipMemset(idx->tags, 0, (size_t)slots * idx->tag_stride));
    HIP_TRY(roar_dev_malloc((void **)&idx->epochs, (size_t)slots * sizeof(uint32_t)));
    HIP_TRY(hipMemset(idx->epochs, 0, (size_t)slots * sizeof(uint32_t)));
    // The fills above run on the null stream, the search on a stream that does not wait for it: without this the kernel marks nodes
    // in lists that are still being zeroed -- marks vanish, nodes are scored again (ids / distances unchanged, cmps above the
    // reference's and different from run to run: seen at 200 k points x 1 000 lists, where the fill takes as long as the launch)
    HIP_TRY(hipDeviceSynchronize());
    idx->slots = slots;
    return ROAR_OK;
}

// Spill lists of the exact LDS form: `slots` lists or none -- a launch with fewer workgroups than the chip holds would cost
// more than the exact cmps count is worth, so when HBM cannot hold them the launch runs the forgetful form instead
// (*granted = false; results are the same, cmps may exceed the reference's).
int ensure_spill(roar_index *idx, uint32_t slots, bool *granted) {
    *granted = true;
    if (idx->spill && idx->spill_slots >= slots) return ROAR_OK;
    const uint64_t stride = std::max<uint64_t>(8, ((idx->n + 7) / 8) * 8);
    if (affordable_slots(stride * 2, idx->spill ? (uint64_t)idx->spill_slots * idx->spill_stride * 2 : 0, slots) < slots) {
        *granted = false;
        return ROAR_OK;
    }
    if (idx->spill) (void)hipFree(idx->spill);
    if (idx->spill_epochs) (void)hipFree(idx->spill_epochs);
    idx->spill = nullptr;
    idx->spill_epochs = nullptr;
    idx->spill_slots = 0;
    idx->spill_stride = stride;
    hipError_t e = roar_dev_malloc((void **)&idx->spill, (size_t)slots * stride * 2);
    if (e == hipSuccess) e = hipMemset(idx->spill, 0, (size_t)slots * stride * 2);
    if (e == hipSuccess) e = roar_dev_malloc((void **)&idx->spill_epochs, (size_t)slots * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(idx->spill_epochs, 0, (size_t)slots * sizeof(uint32_t));
    if (e == hipSuccess) e = hipDeviceSynchronize();  // (see ensure_tags)
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (idx->spill) (void)hipFree(idx->spill);
        if (idx->spill_epochs) (void)hipFree(idx->spill_epochs);
        idx->spill = nullptr;
        idx->spill_epochs = nullptr;
        *granted = false;
        return ROAR_OK;
    }
    idx->spill_slots = slots;
    return ROAR_OK;
}

int ensure_tags32(roar_index *idx, uint32_t *slots_io) {
    uint32_t slots = *slots_io;
    if (idx->tags32 && idx->slots32 >= slots) return ROAR_OK;
    const uint64_t stride = ((idx->n + 3) / 4) * 4;
    slots = affordable_slots(stride * 4, idx->tags32 ? (uint64_t)idx->slots32 * idx->tag_stride32 * 4 : 0, slots);
    *slots_io = slots;
    if (idx->tags32 && idx->slots32 >= slots) return ROAR_OK;
    if (idx->tags32) (void)hipFree(idx->tags32);
    if (idx->epochs32) (void)hipFree(idx->epochs32);
    idx->tags32 = nullptr;
    idx->epochs32 = nullptr;
    idx->slots32 = 0;
    idx->tag_stride32 = stride;
    HIP_TRY(roar_dev_malloc((void **)&idx->tags32, (size_t)slots * idx->tag_stride32 * sizeof(uint32_t)));
    HIP_TRY(hipMemset(idx->tags32, 0, (size_t)slots * idx->tag_stride32 * sizeof(uint32_t)));
    HIP_TRY(roar_dev_malloc((void **)&idx->epochs32, (size_t)slots * sizeof(uint32_t)));
    HIP_TRY(hipMemset(idx->epochs32, 0, (size_t)slots * sizeof(uint32_t)));
    HIP_TRY(hipDeviceSynchronize());  // (see ensure_tags)
    idx->slots32 = slots;
    return ROAR_OK;
}

// Host CSR copy of a graph that was built on the device (download + strip the row padding).
int ensure_host_csr(roar_index *idx) {
    if (!idx->h_off.empty() || idx->gn == 0) return ROAR_OK;
    const uint64_t n = idx->gn;
    const uint32_t S = idx->adj_stride;
    idx->h_off.assign(n + 1, 0);
    idx->h_nbrs.clear();
    const uint64_t chunk = std::max<uint64_t>(1, (128ull << 20) / (S * 4));
    std::vector<uint32_t> buf((size_t)chunk * S);
    for (uint64_t r0 = 0; r0 < n; r0 += chunk) {
        const uint64_t cnt = std::min(chunk, n - r0);
        HIP_TRY(hipMemcpy(buf.data(), idx->adj + r0 * S, (size_t)cnt * S * 4, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < cnt; ++i) {
            const uint32_t *row = buf.data() + i * S;
            idx->h_off[r0 + i] = idx->h_nbrs.size();
            idx->h_nbrs.insert(idx->h_nbrs.end(), row + 1, row + 1 + row[0]);
        }
    }
    idx->h_off[n] = idx->h_nbrs.size();
    idx->nedges = idx->h_nbrs.size();
    return ROAR_OK;
}

constexpr uint32_t LAUNCH_RING = 256;  // launch cells: more launches than this are never in flight on one index
constexpr size_t ADJ_SLACK_WORDS = 256;  // a row read is up to 4 x 64 words from the row's first word

// Hub-annotated adjacency (see RoarSearchParams::adj_hub), built on the device (roar_hub.hip), once per graph: in-degrees,
// the H = min(65 535, n) nodes with the most in-edges (ties to the smaller id) get bitmap slots 0..H-1, every row is
// rewritten as [degree | hubs << 16][hub neighbours, in row order][the others, in row order][u16 slot per hub neighbour].
// Within a hop the scan order cannot change the pool, so listing hubs first changes no result.
int ensure_hub_rows(roar_index *idx) {
    if (idx->hub_state != 0) return ROAR_OK;
    if (!idx->adj || idx->gn == 0) return ROAR_OK;
    if (idx->max_deg > 170 || idx->gn < 2) {
        idx->hub_state = -1;
        return ROAR_OK;
    }
    const uint64_t n = idx->gn;
    const uint32_t Sh = round_up(1 + idx->max_deg + (idx->max_deg + 1) / 2, 32);  // <= 256
    uint32_t *hub = nullptr, *indeg = nullptr;
    uint16_t *slot_of = nullptr;
    unsigned long long *ka = nullptr, *kb = nullptr;
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    auto bail = [&](hipError_t e, const char *what) {  // every failure leaves the index without hub rows, and nothing behind
        (void)hipFree(hub), (void)hipFree(indeg), (void)hipFree(slot_of), (void)hipFree(ka), (void)hipFree(kb), (void)hipFree(tmp);
        return fail(ROAR_ERR_DEVICE, "hub rows: %s: %s", what, hipGetErrorString(e));
    };
    hipError_t e = (hipError_t)roar_hub_build_rows(nullptr, 0, n, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, &tmp_bytes, nullptr);
    if (e != hipSuccess) return bail(e, "sizing the sort");
    const size_t hub_words = (size_t)n * Sh + ADJ_SLACK_WORDS;
    if ((e = roar_dev_malloc((void **)&hub, hub_words * 4)) != hipSuccess) return bail(e, "out of device memory");
    if ((e = roar_dev_malloc((void **)&indeg, (size_t)n * 4)) != hipSuccess) return bail(e, "out of device memory");
    if ((e = roar_dev_malloc((void **)&slot_of, (size_t)n * 2)) != hipSuccess) return bail(e, "out of device memory");
    if ((e = roar_dev_malloc((void **)&ka, (size_t)n * 8)) != hipSuccess) return bail(e, "out of device memory");
    if ((e = roar_dev_malloc((void **)&kb, (size_t)n * 8)) != hipSuccess) return bail(e, "out of device memory");
    if ((e = roar_dev_malloc(&tmp, tmp_bytes)) != hipSuccess) return bail(e, "out of device memory");
    if ((e = hipMemset(hub, 0, hub_words * 4)) != hipSuccess) return bail(e, "memset");
    e = (hipError_t)roar_hub_build_rows(idx->adj, idx->adj_stride, n, hub, Sh, indeg, slot_of, ka, kb, tmp, &tmp_bytes, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return bail(e, "kernels");
    (void)hipFree(indeg), (void)hipFree(slot_of), (void)hipFree(ka), (void)hipFree(kb), (void)hipFree(tmp);
    idx->adj_hub = hub;
    idx->hub_stride = Sh;
    idx->hub_state = 1;
    return ROAR_OK;
}

// The screening copy (roar_search_v2.h, screen8_pass): + n x roar_screen8_stride(dim) bytes of HBM (256 at 200-d, 528 at 512-d), one
// conversion pass.  Failure to make it is not an
// error: the launch then scores every neighbour on the f32 rows, as without it.
void ensure_screen(roar_index *idx) {
    if (idx->screen && idx->screen_n == idx->n) return;
    if (idx->screen) (void)hipFree(idx->screen);
    if (idx->screen_mu) (void)hipFree(idx->screen_mu);
    idx->screen = nullptr;
    idx->screen_mu = nullptr;
    idx->screen_n = 0;
    auto drop = [&]() {
        (void)hipGetLastError();
        if (idx->screen) (void)hipFree(idx->screen);
        if (idx->screen_mu) (void)hipFree(idx->screen_mu);
        idx->screen = nullptr;
        idx->screen_mu = nullptr;
    };
    if (roar_dev_malloc((void **)&idx->screen, (size_t)idx->n * roar_screen8_stride(idx->dim_pad) + 256) != hipSuccess) return drop();
    int rc = 0;
    if (idx->ip) {
        // Centred codes (roar_search_v2.h): the centre is the component-wise MEDIAN of a strided sample of (at most) 4 096 rows -- any
        // finite vector is a valid centre (a bad one only makes the bound loose), the median is one a minority of rows cannot move: a
        // few rows scaled by 1e15, or a thousand rows of large positive integers among Gaussian ones, leave it with the majority
        // (a mean -- first form -- followed them, and the bound of every ordinary row went with it).
        const uint32_t m = (uint32_t)std::min<uint64_t>(idx->n, 4096), dp = idx->dim_pad;
        if (roar_dev_malloc((void **)&idx->screen_mu, (size_t)dp * sizeof(float)) != hipSuccess) return drop();
        std::vector<float> rows, col, mu;
        try {
            rows.resize((size_t)m * dp), col.resize(m), mu.assign(dp, 0.0f);
        } catch (...) {
            return drop();
        }
        const uint64_t every = std::max<uint64_t>(1, idx->n / m);
        if (hipMemcpy2D(rows.data(), (size_t)dp * 4, idx->base, (size_t)every * idx->ld * 4, (size_t)dp * 4, m, hipMemcpyDeviceToHost) != hipSuccess)
            return drop();
        bool finite = true;
        for (uint32_t c = 0; c < dp && finite; ++c) {
            for (uint32_t r = 0; r < m; ++r) {
                const float v = rows[(size_t)r * dp + c];
                col[r] = (v == v) ? v : INFINITY;  // (a NaN sorts as +inf: it moves the median by at most one rank)
            }
            std::nth_element(col.begin(), col.begin() + m / 2, col.end());
            mu[c] = col[m / 2];
            finite = std::fabs(mu[c]) <= 1.0e30f;
        }
        if (!finite) mu.assign(dp, 0.0f);  // (more than half of the sample is not finite: no centre)
        if (hipMemcpy(idx->screen_mu, mu.data(), (size_t)dp * 4, hipMemcpyHostToDevice) != hipSuccess) return drop();
    }
    if (rc == 0) rc = roar_launch_make_screen8(idx->base, idx->ld, idx->dim_pad, idx->n, idx->screen, idx->ip ? 0 : 1, idx->screen_mu, nullptr);
    hipError_t e = rc == 0 ? hipDeviceSynchronize() : (hipError_t)rc;
    if (e != hipSuccess) return drop();
    idx->screen_n = idx->n;
}

// The copy is made when the base rows arrive (one conversion pass), not inside the first search that wants it.
void make_screen_with_base(roar_index *idx) {
    const bool kernel = idx->dim_pad == 128 || idx->dim_pad == 512 || (idx->ip && idx->dim_pad == 200);
    if (kernel && idx->screen_mode != 1 && (idx->screen_mode == 2 || idx->dim_pad * 4u >= 768u)) ensure_screen(idx);
}

struct LaunchPlan {
    RoarSearchParams p;
    RoarKernelSel sel;
    uint32_t grid;
};

// The three things a plan leaves to measurement (roar_index::Calibration); -1 / 0 = not decided: plan_search's defaults.
struct PlanPick {
    int screen = -1;       // 0 off, 1 on
    int mode = -1;         // ROAR_VISITED_EXACT | ROAR_VISITED_LDS (under AUTO: the forgetful form)
    uint32_t entries = 0;  // tags of the on-chip table
    int protect = -1;      // forgetful table of an LDS-pool launch: tags of pool members are never the victim (0 off, 1 on)
};

// does this index have a screening kernel for its rows (roar_search_v2.h: 128-d and 512-d rows, 200-d inner product)
inline bool has_screen_kernel(const roar_index *idx) {
    return idx->dim_pad == 128 || idx->dim_pad == 512 || (idx->ip && idx->dim_pad == 200);
}

// LDS carve-up and grid for one search launch.  `pick`: measured choices for what no option fixes (null: defaults).
int plan_search(roar_index *idx, uint64_t nq, uint32_t k, uint32_t L, LaunchPlan *plan, bool bipartite = false, const PlanPick *pick = nullptr) {
    if (!idx->base || idx->n == 0) return fail(ROAR_ERR_ARG, "search before base vectors were loaded");
    if (bipartite) {
        if (!idx->bp_off) return fail(ROAR_ERR_ARG, "search before the bipartite graph was loaded");
    } else {
        if (!idx->adj) return fail(ROAR_ERR_ARG, "search before the projection graph was loaded");
        if (idx->gn != idx->n)
            return fail(ROAR_ERR_ARG, "graph has %llu nodes but base has %llu rows", (unsigned long long)idx->gn,
                        (unsigned long long)idx->n);
    }
    if (L == 0 || k == 0) return fail(ROAR_ERR_ARG, "k and L_pq must be positive");
    if (k > L) return fail(ROAR_ERR_ARG, "L_pq must greater or equal than k");  // tests/test_search_roargraph.cpp:192-195
    if (idx->n >= (1ull << 31)) return fail(ROAR_ERR_ARG, "more than 2^31 base points are not supported");
    if (nq >= (1ull << 32)) return fail(ROAR_ERR_ARG, "too many queries in one batch");
    RoarSearchParams &p = plan->p;
    memset(&p, 0, sizeof p);
    p.base = idx->base;
    p.ld = idx->ld;
    p.dim = idx->dim_pad;
    idx->last_screen = 0;
    // The 8-bit screen: option "screen" 1 = off, 2 = on; AUTO: what the calibration measured for this kind of launch, else (no
    // calibration: option, builder's searches) on for rows of 768 bytes and more -- at 128-d a hop gathers too few bytes for the
    // second dependent round trip to pay on any set measured (SIFT-1M shape, off / on: L_pq = 170 6.78 / 12.8 ms).
    {
        bool want_screen = false;
        if (!bipartite && has_screen_kernel(idx) && idx->screen_mode != 1) {
            if (idx->screen_mode == 2) want_screen = true;
            else if (pick && pick->screen >= 0 && !idx->building) want_screen = pick->screen != 0;
            else want_screen = idx->dim_pad * 4u >= 768u;
        }
        if (want_screen) {
            ensure_screen(idx);
            p.screen = idx->screen;
            p.screen_mu = idx->screen ? idx->screen_mu : nullptr;
            idx->last_screen = idx->screen ? 1 : 0;
        }
    }
    p.adj = idx->adj;
    p.adj_stride = idx->adj_stride;
    p.n = (uint32_t)idx->n;
    p.ep = idx->ep;
    p.nq = (uint32_t)nq;
    p.k = k;
    p.L = L;
    // Hub-bitmap form (long pools on graphs whose hops gather many bytes): the 65 535 highest-in-degree nodes keep their visited
    // bits in an exact LDS bitmap, the tag table / the HBM tags see only the other neighbours.  It needs a second copy of the
    // adjacency (seconds to make, once per graph), so it is not a calibration candidate but a structural rule: "dense" =
    // average degree x row bytes >= 24 x 800 B per hop, without the screen (with it a hop gathers the 8-bit rows of its
    // neighbours and the f32 rows of the few that pass: every graph measured then behaves like a sparse one).  T2I-10M bootstrap
    // graph (average degree 33, ~97 rows per hop), 10 000 queries, best plain form -> best hub form: L_pq = 100 9.96 -> 9.91 ms,
    // 500 37.60 -> 36.85, 1 000 85.7 -> 68.5, 2 000 142.2 -> 128.6 ms (profiles/r02_hub_forms*.txt); a 128-d graph of average
    // degree 32 gathers 16 KB per hop and behaves like a sparse one (SIFT-1M shape, L_pq = 170: hub form 9.59 ms, plain 8.0 ms).
    const bool dense = !p.screen && idx->gn && (double)idx->nedges / (double)idx->gn * (double)idx->dim_pad * 4.0 >= 24.0 * 800.0;
    bool use_hub = false;
    if (!bipartite && !idx->building && L > 64 && idx->hub_mode != 1 && (idx->hub_mode == 2 || (dense && L >= 100))) {
        int e = ensure_hub_rows(idx);
        if (e) return e;
        use_hub = idx->hub_state == 1;
    }
    // Visited form.  An option fixes it; under AUTO it is what the calibration measured on this index (the on-chip tag table,
    // forgetful, against u8 epoch tags in HBM: the table wins while the pool's nodes mostly fit it, the HBM tags beyond -- where
    // "beyond" starts depends on the graph, the row length and the screen: L_pq ~ 500 ... 1 950 on the sets measured so far,
    // profiles/r02_hub_forms*.txt, r03_screen_forms.txt, r04_long_pools_lds.txt, r04_bench_L1000_stderr.txt).  Without a measurement (option "calibrate" = 0,
    // the builder's searches): the table up to L_pq = 1 536.
    // (a u16 tag needs sets >= n / 65534; beyond ~134 M points per index that table no longer fits the cap)
    const bool lds_fits = (idx->n + 65533) / 65534 <= 2048;
    // (the two-hop walk touches an order of magnitude more distinct nodes per hop: the 8192-tag table forgets
    // so much that LDS mode re-scores 1.5-2.7x the nodes, T2I-10M bipartite sweep: EXACT 21.5 / 47.0 / 94.0 ms
    // vs LDS 34.0 / 79.7 / 185 ms at L_pq = 10 / 100 / 500 -> AUTO is EXACT there)
    int mode = idx->visited_mode;
    if (mode == ROAR_VISITED_AUTO) {
        if (bipartite || !lds_fits) mode = ROAR_VISITED_EXACT;
        else if (pick && pick->mode >= 0 && !idx->building) mode = pick->mode == ROAR_VISITED_EXACT ? ROAR_VISITED_EXACT : ROAR_VISITED_LDS;
        else mode = L < 1536u ? ROAR_VISITED_LDS : ROAR_VISITED_EXACT;
    }
    // LDS: the exact form (ids that find their set full go to the spill list in HBM); LDS_FORGETFUL: a full set overwrites
    // a victim (cmps >= the reference's, nothing stored to HBM); the builder's searches and the two-hop walk forget
    // AUTO takes the forgetful form wherever it takes the table at all: it is the faster one at every L_pq on both benchmark
    // graphs (T2I-10M, 10 000 queries, forgetful / exact: L_pq = 55 3.652 / 3.667 ms, 100 6.02 / 6.29, 200 11.04 / 12.00, 500
    // 25.8 / 30.1, 1 000 54.1 / 61.6, 1 400 77.0 / 84.2, 2 000 117.8 / 125.2 ms; profiles/r03_visited_forms.txt) -- a one-word
    // mark stored to HBM costs as much as 2-3.5 lines of a row gather (profiles/r03_probe_cost.txt), an id beyond the
    // table's capacity is asked for again only one time in three, and a forgotten one costs its 7 row lines only then.
    bool forget = mode == ROAR_VISITED_LDS_FORGETFUL || idx->building || bipartite ||
                  (idx->visited_mode == ROAR_VISITED_AUTO && mode == ROAR_VISITED_LDS);
    if (mode == ROAR_VISITED_LDS_FORGETFUL) mode = ROAR_VISITED_LDS;
    // bipartite, AUTO: eight waves per query while the pool is short (work per query is far too uneven for one
    // wave each: T2I-10M, 10 000 queries, 14.0 / 20.0 / 32.8 / 44.9 / 68.2 ms at L_pq = 10 / 20 / 50 / 100 / 200 against
    // 20.7 / 26.4 / 39.4 / 46.0 / 68.0 ms with one wave; from L_pq = 500 the waves queue for the shared pool and the
    // one-wave form wins, 99 vs 113 ms); an explicit visited_mode keeps the one-wave form with that mode
    const bool bp_wg = bipartite && idx->visited_mode == ROAR_VISITED_AUTO && L <= 200;
    idx->last_mode = mode;
    // visited table size: sets (2^k or 3 * 2^k) >= n / 65535 so that the u16 tag is exact.  It trades forgetting against resident
    // workgroups -- which way pays depends on the graph, the row length, L_pq and the batch: option "vis_entries", else what the
    // calibration measured, else 8 192 tags (4 096 with the screen or with rows under 768 bytes: such launches live on resident
    // workgroups).
    uint32_t sets_log2 = 0;
    uint32_t entries = 0;
    bool sets3 = false;
    if (mode == ROAR_VISITED_LDS) {
        uint32_t want = (p.screen || idx->dim_pad * 4u < 768u) ? 4096u : 8192u;
        if (pick && pick->entries > 0 && !idx->building && !bipartite) want = pick->entries;
        if (idx->vis_entries > 0) want = (uint32_t)idx->vis_entries;
        uint32_t min_sets = (uint32_t)((idx->n + 65533) / 65534);
        uint32_t sets = std::max(std::max(want / 4, min_sets), 16u);
        while ((1u << sets_log2) < sets) sets_log2++;
        // 3 * 2^k sets (12 288, 24 576 ... tags) where that is what was asked for: the search kernel only
        if (!bipartite && sets_log2 >= 2 && sets <= (3u << (sets_log2 - 2)) && (3u << (sets_log2 - 2)) >= min_sets) {
            sets_log2 -= 2;
            sets3 = true;
        }
        entries = (sets3 ? 12u : 4u) << sets_log2;
        if ((uint64_t)entries * 2 > 128 * 1024)
            return fail(ROAR_ERR_ARG, "visited table of %u entries does not fit in LDS", entries);
    }
    p.vis_sets_log2 = sets_log2 | (sets3 ? 0x80000000u : 0u);  // (+ bit 30 once the spill lists are known to fit, below)
    uint32_t off = 0;
    p.off_pool = off;
    off += round_up(8 * (L + 1), 16);
    // candidate ids / keys of one adjacency round trip: three 64-word chunks of a row -- two when no row has a third (stride <= 128 words:
    // the benchmark graphs), one for stride 64.  768 bytes less per query is a workgroup more per CU at every other L_pq of a long pool
    // (the LDS is handed out in 1 280-byte blocks, below), and long pools live on resident workgroups: T2I-10M, L_pq = 1 000, 2 048 tags,
    // nine per CU 36.2 ms -> ten per CU (profiles/r04_long_pools_lds.txt).
    const uint32_t trip = (use_hub || bipartite || idx->adj_stride > 128u) ? 192u : idx->adj_stride > 64u ? 128u : 64u;
    p.off_ck = off;
    off += bp_wg ? 8 * 64 * 8 : 8 * trip;   // eight waves x 64 keys in the workgroup form; else one adjacency round trip
    p.off_cid = off;
    off += bp_wg ? 4 * 64 * 8 : 4 * trip;
    p.off_q = off;
    off += round_up(4 * idx->dim_pad, 16);
    if (p.screen) off += round_up(idx->dim_pad, 16);  // the query's 8-bit codes (screen8i_pass)
    p.off_vis = off;
    off += 2 * entries;
    p.off_ctrl = off;
    off += 64;  // (words 0-5 multi-wave hand-over, 6 spill epoch, 8-10 the query's side of the screen, 12-13 traffic counters)
    if (use_hub) {  // row buffer + bitmap of the hub form
        p.off_rowbuf = off;
        off += 1024;
        p.off_hbits = off;
        off += 8192;
        p.adj_hub = idx->adj_hub;
        p.hub_stride = idx->hub_stride;
    }
    if (bipartite) {
        p.off_bp = off;
        off += 768;
        p.bp_off = idx->bp_off;
        p.bp_nbrs = idx->bp_nbrs;
    }
    p.lds_bytes = off;
    if (p.lds_bytes > 160 * 1024) return fail(ROAR_ERR_ARG, "L_pq = %u needs %u bytes of LDS per query (max 163840)", L, off);
    // kernel form: W waves per query while the whole batch is resident that way (latency forms), one wave per
    // query otherwise (throughput form)
    RoarKernelSel &sel = plan->sel;
    memset(&sel, 0, sizeof sel);
    sel.metric_ip = idx->ip ? 1 : 0;
    sel.visited_mode = mode;
    sel.form = ROAR_FORM_SEARCH;
    sel.waves = 1;
    sel.regpool = (L <= 64 && idx->pool_form == 0) ? 1 : 0;
    sel.prof = idx->profile ? 1 : 0;
    sel.hub = use_hub ? 1 : 0;
    if (sel.prof && !(idx->ip && idx->dim_pad == 200) && !(!idx->ip && idx->dim_pad == 128))
        return fail(ROAR_ERR_ARG, "profile_phases: the instrumented kernels exist for the 200-d inner-product and the 128-d L2 shapes only");
    int occ = 0;
    auto occupancy = [&](int *out) -> int {
        // (asked once per kernel form and LDS size: a stream of launches of one shape must not pay a runtime query each)
        const uint64_t key = ((uint64_t)p.lds_bytes << 32) | ((uint64_t)(sel.form & 7) << 28) | ((uint64_t)(sel.waves & 7) << 24) |
                             ((uint64_t)(sel.visited_mode & 3) << 20) | ((uint64_t)(sel.regpool & 1) << 19) | ((uint64_t)(sel.prof & 1) << 18) |
                             ((uint64_t)(sel.hub & 1) << 17) | ((uint64_t)(sel.metric_ip & 1) << 16) | (uint64_t)(idx->dim_pad & 0xFFFF);
        for (const auto &e : idx->occ_cache)
            if (e.first == key) {
                *out = e.second;
                return ROAR_OK;
            }
        int rc = roar_search_occupancy(&sel, idx->dim_pad, p.lds_bytes, out);
        if (rc != 0) return fail(ROAR_ERR_DEVICE, "occupancy query failed: %s", hipGetErrorString((hipError_t)rc));
        if (idx->occ_cache.size() < 64) idx->occ_cache.emplace_back(key, *out);
        return ROAR_OK;
    };
    bool picked = false;
    if (bipartite) {
        sel.form = bp_wg ? ROAR_FORM_BIPARTITE_WG : ROAR_FORM_BIPARTITE;
        sel.prof = sel.regpool = sel.hub = 0;
        int e = occupancy(&occ);
        if (e) return e;
        picked = true;
    } else if (idx->building && mode == ROAR_VISITED_LDS) {
        sel.form = ROAR_FORM_BUILDER;
        sel.prof = sel.hub = 0;
        int e = occupancy(&occ);
        if (e) return e;
        picked = true;
    } else if (idx->search_kernel != 1 && !idx->building) {
        // search_kernel: 0 auto, 1 one wave per query, 2 four waves, 3 two waves.  Auto: four waves for batches of at most
        // one query per CU on a dense graph (a hop gathers ~100 rows: one pass instead of three -- 256 queries at L_pq = 30
        // on the T2I-10M bootstrap graph 0.305 vs 0.329 ms); one wave otherwise.  On a sparse graph a hop gathers ~50 rows
        // and the extra waves only add barriers (256 queries at L_pq = 55: 0.353 ms with one wave, 0.369 with four), and from a
        // few hundred queries on the one-wave form is ahead everywhere (1 250 queries: 0.549 vs 0.557 ms with two waves).
        // (With the 8-bit screen `dense` is false: the extra waves share the exact scoring pass, and the screen leaves it few rows.)
        for (int w : {4, 2}) {
            if (idx->search_kernel == 2 && w != 4) continue;
            if (idx->search_kernel == 3 && w != 2) continue;
            if (idx->search_kernel == 0 && (w != 4 || !dense || nq > (uint64_t)idx->n_cus)) continue;
            sel.waves = w;
            if (sel.prof && sel.hub) continue;  // (the instrumented hub form exists for one wave per query)
            int occ_mw = 0;
            int e = occupancy(&occ_mw);
            if (e) return e;
            if (occ_mw >= 1) {
                occ = occ_mw;
                picked = true;
                break;
            }
        }
    }
    if (!picked) {
        sel.waves = 1;
        int e = occupancy(&occ);
        if (e) return e;
    }
    idx->last_sel = sel;
    // The LDS of a gfx950 CU is handed out in 1 280-byte blocks (160 KB / 128): a 32 128-byte workgroup takes 33 280 and
    // four of them fit, not the five the occupancy query reports (measured: the 10 000-query launch at L_pq = 550 with
    // 32 128 B ran at 4 per CU speed, 31 728 B at L_pq = 500 at 5 per CU speed).
    {
        const uint32_t alloc = round_up(std::max<uint32_t>(p.lds_bytes, 1u), 1280u);
        const int by_lds = (int)((160u * 1024u) / alloc);  // workgroups per CU
        if (by_lds >= 1 && occ > by_lds) occ = by_lds;
    }
    if (occ < 1) occ = 1;
    int wpc = idx->waves_per_cu > 0 ? std::min(idx->waves_per_cu, occ) : occ;
    // (Rounds 2-3 launched six one-wave workgroups per CU with the on-chip table and no screen although the LDS allows eight: 1 % on the
    // 10 M-point benchmark -- 3.593 against 3.635 ms -- and a handicap wherever a launch is bound by latency, e.g. any set that fits the
    // caches; the cap went with the other constants measured on one graph.  Option "waves_per_cu" still sets it.)
    uint64_t grid = (uint64_t)wpc * (uint64_t)idx->n_cus;
    if (grid > nq) grid = nq;
    if (idx->max_grid > 0 && grid > (uint64_t)idx->max_grid) grid = idx->max_grid;
    if (grid == 0) grid = 1;
    plan->grid = (uint32_t)grid;
    if (bp_wg) {
        // u32 tags: 4 n bytes per resident workgroup; 256 workgroups keep every CU busy with eight waves each
        if (plan->grid > 256) plan->grid = 256;
        uint32_t slots = plan->grid;
        int e = ensure_tags32(idx, &slots);
        if (e) return e;
        plan->grid = std::min(plan->grid, std::max(slots, idx->slots32));
        p.tags32 = idx->tags32;
        p.tag_stride32 = idx->tag_stride32;
        p.epochs = idx->epochs32;
    } else if (mode == ROAR_VISITED_EXACT) {
        uint32_t slots = plan->grid;
        int e = ensure_tags(idx, &slots);
        if (e) return e;
        plan->grid = std::min(plan->grid, std::max(slots, idx->slots));
        p.tags = idx->tags;
        p.tag_stride = idx->tag_stride;
        p.epochs = idx->epochs;
    } else if (!forget) {
        bool granted = false;
        int e = ensure_spill(idx, plan->grid, &granted);
        if (e) return e;
        if (granted) {
            p.tags = reinterpret_cast<uint8_t *>(idx->spill);
            p.tag_stride = idx->spill_stride;
            p.epochs = idx->spill_epochs;
        } else {
            forget = true;
        }
    }
    // (the two-hop kernels always forget and read the field as a plain shift count: the flag is the search kernel's)
    if (mode == ROAR_VISITED_LDS && forget && !bipartite) p.vis_sets_log2 |= 0x40000000u;
    // bit 29: bit 15 of a tag says "its node is in the pool" and the victim choice skips such ways (LDS pool only; every tag must stay
    // below 2^15: base points / sets < 32 767)
    // Whether that pays is the price of a revisit against ~60 more instructions per hop: with the screen a forgotten pool member costs
    // an 8-bit row, an f32 row and a merge candidate (T2I-10M, L_pq = 1 000: 32.2 -> 30.2 ms, 500: 15.1 -> 13.6), without it one row
    // (SIFT shape, L_pq = 170: 6.34 -> 6.53 ms).  Option "tag_protect", else what the calibration measured, else on (cmps closer to the
    // reference's, and no dependence of cmps on the screen).
    idx->last_tag_protect = 0;
    const int want_protect = idx->tag_protect >= 0 ? idx->tag_protect : (pick && pick->protect >= 0) ? pick->protect : 1;
    sel.protect = 0;
    if (mode == ROAR_VISITED_LDS && forget && !bipartite && sel.form == ROAR_FORM_SEARCH && !sel.regpool && sel.waves == 1 && !sel.hub &&
        want_protect) {
        const uint64_t sets = (uint64_t)(sets3 ? 3u : 1u) << sets_log2;
        if (idx->n > 0 && (idx->n - 1) / sets + 1 < 32768u) {
            p.vis_sets_log2 |= 0x20000000u;
            sel.protect = 1;
            idx->last_tag_protect = 1;
        }
    }
    idx->last_sel = sel;
    idx->last_forgetful = (mode == ROAR_VISITED_LDS && forget) ? 1 : 0;
    p.counter = idx->counters;
    if (sel.prof) {
        if (idx->prof_cap < nq) {
            if (idx->prof_dev) (void)hipFree(idx->prof_dev);
            idx->prof_dev = nullptr;
            idx->prof_cap = 0;
            HIP_TRY(roar_dev_malloc((void **)&idx->prof_dev, (size_t)nq * 16 * sizeof(unsigned long long)));
            idx->prof_cap = nq;
        }
        p.prof = idx->prof_dev;
        idx->prof_nq = nq;
    }
    idx->last_grid = plan->grid;
    idx->last_lds = p.lds_bytes;
    idx->last_vis_entries = entries;
    return ROAR_OK;
}

// Host-buffer queries -> something the kernel can read.  Rows are padded to dim_pad with zeros (data_align of the
// query block, tests/test_search_roargraph.cpp:131).  Two ways, neither with a host-to-device copy in front of the
// kernel: (1) the caller's block is already pinned (roar_host_alloc, hipHostMalloc, hipHostRegister), strided at
// >= dim_pad floats with nothing to zero -> the kernel reads it in place over PCIe (each query is read once, 4 * dim
// bytes, at its own start: 8 MB per 10 000 queries, hidden under the search); (2) otherwise the rows are packed into
// the index's pinned staging block by a few host threads and read from there.  *qdev / *qstride: what to launch with.
int stage_queries(roar_index *idx, const float *queries, uint64_t nq, uint32_t qstride_floats, uint32_t k, const float **qdev,
                  uint32_t *qstride, bool *deferred) {
    *deferred = false;
    const uint64_t owords = nq * (2ull * k + 3) + 1;
    if (idx->out_cap < owords) {
        if (idx->out_dev) (void)hipFree(idx->out_dev);
        if (idx->out_pin) (void)hipHostFree(idx->out_pin);
        idx->out_dev = idx->out_pin = nullptr;
        idx->out_cap = 0;
        HIP_TRY(roar_dev_malloc((void **)&idx->out_dev, (size_t)owords * sizeof(uint32_t)));
        HIP_TRY(hipHostMalloc((void **)&idx->out_pin, (size_t)owords * sizeof(uint32_t), hipHostMallocDefault));
        idx->out_cap = owords;
    }
    if (idx->dim == idx->dim_pad && qstride_floats >= idx->dim_pad && (qstride_floats & 3) == 0) {
        hipPointerAttribute_t attr;
        memset(&attr, 0, sizeof attr);
        if (hipPointerGetAttributes(&attr, queries) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer) {
            *qdev = static_cast<const float *>(attr.devicePointer);
            *qstride = qstride_floats;
            return ROAR_OK;
        }
        (void)hipGetLastError();  // an ordinary (pageable) pointer is reported as an error by some runtimes
    }
    // staging rows start on 128-byte lines: the kernel reads rows while later ones are still being written
    // (pack_queries), and no line it fetches may hold bytes of a row that is not there yet
    const uint32_t sp = round_up(idx->dim_pad, 32);
    const uint64_t qfloats = nq * sp;
    if (idx->q_cap < qfloats) {
        if (idx->q_pin) (void)hipHostFree(idx->q_pin);
        idx->q_pin = idx->q_pin_dev = nullptr;
        idx->q_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&idx->q_pin, (size_t)qfloats * sizeof(float), hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void **)&idx->q_pin_dev, idx->q_pin, 0));
        idx->q_cap = qfloats;
    }
    if (!idx->ready_pin) {
        HIP_TRY(hipHostMalloc((void **)&idx->ready_pin, 64, hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void **)&idx->ready_dev, idx->ready_pin, 0));
    }
    *idx->ready_pin = 0;  // nothing packed yet: pack_queries() runs behind the launch
    *qdev = idx->q_pin_dev;
    *qstride = sp;
    *deferred = true;
    return ROAR_OK;
}

// The packing itself, on the calling thread, while the kernel is already running: the kernel takes queries in row order
// and waits for the counter (RoarSearchParams::ready), so the 8 MB copy of a 10 000-query block costs the launch nothing
// but the first few rows (two helper threads claiming chunks in order changed nothing: 2.51 vs 2.54 M QPS on the T2I-10M
// bench).  Plain memcpy / memset: nothing here can throw or fail, the counter always reaches nq.
void pack_queries(roar_index *idx, const float *queries, uint64_t nq, uint32_t qstride_floats) {
    const uint32_t dp = round_up(idx->dim_pad, 32), d = idx->dim;  // (the stride stage_queries chose)
    volatile uint32_t *ready = idx->ready_pin;
    for (uint64_t r0 = 0; r0 < nq;) {
        const uint64_t r1 = std::min<uint64_t>(nq, r0 + (r0 < 4096 ? 256 : 1024));  // small steps first: the launch is waiting
        for (uint64_t i = r0; i < r1; ++i) {
            float *dst = idx->q_pin + i * dp;
            memcpy(dst, queries + i * qstride_floats, d * sizeof(float));
            if (dp > d) memset(dst + d, 0, (dp - d) * sizeof(float));  // anything past dim is padding and must read as zero
        }
        std::atomic_thread_fence(std::memory_order_release);
        *ready = (uint32_t)r1;
        r0 = r1;
    }
}

// results: one asynchronous copy of the block [ids | dists | cmps | hops | scanned | short] into pinned memory,
// then plain copies into the caller's arrays
int fetch_results(roar_index *idx, uint64_t nq, uint32_t k, uint32_t *ids, float *dists, uint32_t *cmps, uint32_t *hops,
                  uint32_t *scanned) {
    const uint64_t words = nq * (2ull * k + 3) + 1;
    HIP_TRY(hipMemcpyAsync(idx->out_pin, idx->out_dev, (size_t)words * 4, hipMemcpyDeviceToHost, idx->stream));
    HIP_TRY(hipStreamSynchronize(idx->stream));
    const uint32_t *blk = idx->out_pin;
    memcpy(ids, blk, (size_t)nq * k * 4);
    memcpy(dists, blk + nq * k, (size_t)nq * k * 4);
    if (cmps) memcpy(cmps, blk + 2 * nq * k, (size_t)nq * 4);
    if (hops) memcpy(hops, blk + 2 * nq * k + nq, (size_t)nq * 4);
    if (scanned) memcpy(scanned, blk + 2 * nq * k + 2 * nq, (size_t)nq * 4);
    const uint32_t nshort = blk[(size_t)words - 1];
    if (nshort) {
        // message of the first short query, as the reference would have thrown it (:2408-2412)
        for (uint64_t i = 0; i < nq; ++i) {
            uint32_t have = 0;
            while (have < k && ids[i * k + have] != 0xFFFFFFFFu) have++;
            if (have < k) return fail(ROAR_ERR_SHORT, "not enough results: %u, expected: %u", have, k);
        }
        return fail(ROAR_ERR_SHORT, "not enough results");
    }
    return ROAR_OK;
}

}  // namespace

extern "C" {

const char *roar_last_error(void) { return g_err.c_str(); }

// (internal, for the library's other translation units: roar_sharded.cpp)
void roar_set_last_error(const char *msg) {
    try {
        g_err = msg ? msg : "";
    } catch (...) {
    }
}

int roar_create(roar_index **out, uint32_t dim, int metric, int device) {
    DeviceRestore keep_device_;
    if (!out) return fail(ROAR_ERR_ARG, "null output pointer");
    *out = nullptr;
    if (dim == 0) return fail(ROAR_ERR_ARG, "dimension must be positive");
    if (metric != ROAR_METRIC_L2 && metric != ROAR_METRIC_IP && metric != ROAR_METRIC_COSINE)
        return fail(ROAR_ERR_ARG, "unknown metric %d", metric);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(ROAR_ERR_DEVICE, "no HIP device available (%s); this library has no CPU path",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(ROAR_ERR_ARG, "device %d out of range (%d devices)", device, count);
    roar_index *idx = new (std::nothrow) roar_index();
    if (!idx) return fail(ROAR_ERR_DEVICE, "out of host memory");
    idx->device = device;
    idx->dim = dim;
    idx->dim_pad = round_up(dim, 8);  // DATA_ALIGN_FACTOR, util.h:38-48
    idx->metric = metric;
    idx->ip = metric != ROAR_METRIC_L2;          // src/index.cpp:11-24
    idx->need_normalize = metric == ROAR_METRIC_COSINE;  // src/index_bipartite.cpp:35-37
    auto init = [&]() -> int {
        HIP_TRY(hipSetDevice(device));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        idx->n_cus = prop.multiProcessorCount;
        HIP_TRY(roar_dev_malloc((void **)&idx->counters, 2 * sizeof(uint32_t)));
        HIP_TRY(roar_dev_malloc((void **)&idx->ring, LAUNCH_RING * 4 * sizeof(uint32_t)));
        HIP_TRY(hipMemset(idx->ring, 0, LAUNCH_RING * 4 * sizeof(uint32_t)));
        HIP_TRY(hipStreamCreateWithFlags(&idx->stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreate(&idx->ev0));
        HIP_TRY(hipEventCreate(&idx->ev1));
        return ROAR_OK;
    };
    const int e2 = init();
    if (e2) {
        const std::string keep = g_err;
        roar_destroy(idx);
        g_err = keep;
        return e2;
    }
    *out = idx;
    return ROAR_OK;
}

void roar_destroy(roar_index *idx) {
    DeviceRestore keep_device_;
    if (!idx) return;
    (void)hipSetDevice(idx->device);
    free_graph(idx);
    free_bipartite(idx);
    free_base(idx);
    free_tags(idx);
    if (idx->init_dev) (void)hipFree(idx->init_dev);
    if (idx->counters) (void)hipFree(idx->counters);
    if (idx->ring) (void)hipFree(idx->ring);
    if (idx->q_pin) (void)hipHostFree(idx->q_pin);
    if (idx->ready_pin) (void)hipHostFree(idx->ready_pin);
    if (idx->out_dev) (void)hipFree(idx->out_dev);
    if (idx->out_pin) (void)hipHostFree(idx->out_pin);
    if (idx->stream) (void)hipStreamDestroy(idx->stream);
    if (idx->prof_dev) (void)hipFree(idx->prof_dev);
    if (idx->ev0) (void)hipEventDestroy(idx->ev0);
    if (idx->ev1) (void)hipEventDestroy(idx->ev1);
    delete idx;
}

int roar_set_base(roar_index *idx, const float *rows, uint64_t n, uint32_t row_stride_floats) {
    DeviceRestore keep_device_;
    if (!idx || !rows || n == 0) return fail(ROAR_ERR_ARG, "bad base arguments");
    if (row_stride_floats < idx->dim) return fail(ROAR_ERR_ARG, "row stride smaller than the dimension");
    int e = use_device(idx);
    if (e) return e;
    if ((e = alloc_base(idx, n))) return e;
    const uint64_t chunk = std::max<uint64_t>(1, (64ull << 20) / (idx->ld * sizeof(float)));
    std::vector<float> buf((size_t)chunk * idx->ld);
    for (uint64_t r0 = 0; r0 < n; r0 += chunk) {
        uint64_t cnt = std::min(chunk, n - r0);
        for (uint64_t i = 0; i < cnt; ++i) {
            float *dst = buf.data() + i * idx->ld;
            memcpy(dst, rows + (r0 + i) * row_stride_floats, idx->dim * sizeof(float));
            memset(dst + idx->dim, 0, (idx->ld - idx->dim) * sizeof(float));
            if (idx->need_normalize) normalize_row(dst, idx->dim);
        }
        if ((e = upload_rows(idx, buf.data(), r0, cnt))) return e;
    }
    make_screen_with_base(idx);
    return ROAR_OK;
}

int roar_set_base_device(roar_index *idx, const float *rows_dev, uint64_t n, uint32_t row_stride_floats) {
    DeviceRestore keep_device_;
    if (!idx || !rows_dev || n == 0) return fail(ROAR_ERR_ARG, "bad base arguments");
    if (row_stride_floats < idx->dim_pad || (row_stride_floats & 3))
        return fail(ROAR_ERR_ARG, "device rows need a stride >= %u floats that is a multiple of 4", idx->dim_pad);
    if (idx->need_normalize) return fail(ROAR_ERR_ARG, "COSINE rows must be normalised by the caller before adoption");
    int e = use_device(idx);
    if (e) return e;
    free_base(idx);
    free_tags(idx);
    idx->base = const_cast<float *>(rows_dev);
    idx->base_owned = false;
    idx->n = n;
    idx->ld = row_stride_floats;
    make_screen_with_base(idx);
    return ROAR_OK;
}

int roar_load_base_fbin(roar_index *idx, const char *path) {
    DeviceRestore keep_device_;
    if (!idx || !path) return fail(ROAR_ERR_ARG, "bad arguments");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(ROAR_ERR_IO, "open file error");  // util.h:109-112
    uint32_t hdr[2];
    if (fread(hdr, 4, 2, f) != 2) {
        fclose(f);
        return fail(ROAR_ERR_FORMAT, "Data file size wrong!");
    }
    fseek(f, 0, SEEK_END);
    const uint64_t fsize = (uint64_t)ftell(f);
    const uint32_t n = hdr[0], d = hdr[1];
    // load_meta: (fsize - 8) / d / 4 must equal n (util.h:118-125)
    if (d == 0 || (uint32_t)((fsize - 8) / d / 4) != n) {
        fclose(f);
        return fail(ROAR_ERR_FORMAT, "Data file size wrong!");
    }
    if (d != idx->dim) {
        fclose(f);
        return fail(ROAR_ERR_ARG, "base and query dimension mismatch");  // src/index_bipartite.cpp:2667-2669
    }
    int e = use_device(idx);
    if (!e) e = alloc_base(idx, n);
    if (e) {
        fclose(f);
        return e;
    }
    fseek(f, 8, SEEK_SET);
    const uint64_t chunk = std::max<uint64_t>(1, (64ull << 20) / (idx->ld * sizeof(float)));
    std::vector<float> buf((size_t)chunk * idx->ld);
    for (uint64_t r0 = 0; r0 < n; r0 += chunk) {
        uint64_t cnt = std::min<uint64_t>(chunk, n - r0);
        for (uint64_t i = 0; i < cnt; ++i) {  // load_data, util.h:196-200
            float *dst = buf.data() + i * idx->ld;
            if (fread(dst, 4, d, f) != d) {
                fclose(f);
                return fail(ROAR_ERR_FORMAT, "Data file size wrong!");
            }
            memset(dst + d, 0, (idx->ld - d) * sizeof(float));
            if (idx->need_normalize) normalize_row(dst, d);
        }
        if ((e = upload_rows(idx, buf.data(), r0, cnt))) {
            fclose(f);
            return e;
        }
    }
    fclose(f);
    make_screen_with_base(idx);
    return ROAR_OK;
}

int roar_set_graph_csr(roar_index *idx, uint32_t ep, uint64_t n, const uint64_t *offsets, const uint32_t *nbrs) {
    DeviceRestore keep_device_;
    if (!idx || !offsets || (n && !nbrs && offsets[n] != 0)) return fail(ROAR_ERR_ARG, "bad graph arguments");
    if (n == 0 || n >= (1ull << 31)) return fail(ROAR_ERR_ARG, "graph must have between 1 and 2^31 nodes");
    if (ep >= n) return fail(ROAR_ERR_FORMAT, "entry point %u outside the graph (%llu nodes)", ep, (unsigned long long)n);
    int e = use_device(idx);
    if (e) return e;
    uint32_t max_deg = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (offsets[i + 1] < offsets[i]) return fail(ROAR_ERR_FORMAT, "CSR offsets are not monotone");
        uint64_t d = offsets[i + 1] - offsets[i];
        if (d > 0xFFFFFFu) return fail(ROAR_ERR_FORMAT, "node %llu has an implausible degree", (unsigned long long)i);
        max_deg = std::max<uint32_t>(max_deg, (uint32_t)d);
    }
    const uint64_t ne = offsets[n];
    for (uint64_t j = 0; j < ne; ++j)
        if (nbrs[j] >= n) return fail(ROAR_ERR_FORMAT, "neighbour id %u outside the graph", nbrs[j]);
    free_graph(idx);
    idx->h_off.assign(offsets, offsets + n + 1);
    idx->h_nbrs.assign(nbrs, nbrs + ne);
    idx->ep = ep;
    idx->gn = n;
    idx->nedges = ne;
    idx->max_deg = max_deg;
    // rows start on 128-byte lines; 64 words of slack so that a 64-lane chunk read never leaves the buffer
    idx->adj_stride = round_up(max_deg + 1, 32);
    const size_t words = (size_t)n * idx->adj_stride + ADJ_SLACK_WORDS;
    HIP_TRY(roar_dev_malloc((void **)&idx->adj, words * sizeof(uint32_t)));
    // Build rows in chunks.  A neighbour repeated inside one list is kept once (first occurrence):
    // the reference skips the repeat through its visited test (index_bipartite.cpp:2378), so the
    // search result and cmps are unchanged, and the device visited filter sees distinct ids per row.
    std::vector<uint32_t> stamp((size_t)n, 0xFFFFFFFFu);
    const uint64_t chunk = std::max<uint64_t>(1, (64ull << 20) / (idx->adj_stride * 4));
    std::vector<uint32_t> buf((size_t)chunk * idx->adj_stride);
    for (uint64_t r0 = 0; r0 < n; r0 += chunk) {
        uint64_t cnt = std::min(chunk, n - r0);
        std::fill(buf.begin(), buf.begin() + cnt * idx->adj_stride, 0u);
        for (uint64_t i = 0; i < cnt; ++i) {
            uint32_t *row = buf.data() + i * idx->adj_stride;
            uint32_t deg = 0;
            for (uint64_t j = offsets[r0 + i]; j < offsets[r0 + i + 1]; ++j) {
                uint32_t nb = nbrs[j];
                if (stamp[nb] == (uint32_t)(r0 + i)) continue;
                stamp[nb] = (uint32_t)(r0 + i);
                row[1 + deg++] = nb;
            }
            row[0] = deg;
        }
        HIP_TRY(hipMemcpy(idx->adj + r0 * idx->adj_stride, buf.data(), (size_t)cnt * idx->adj_stride * 4,
                          hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemset(idx->adj + (size_t)n * idx->adj_stride, 0, ADJ_SLACK_WORDS * sizeof(uint32_t)));
    return ROAR_OK;
}

int roar_load_graph_file(roar_index *idx, const char *path) {
    if (!idx || !path) return fail(ROAR_ERR_ARG, "bad arguments");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(ROAR_ERR_IO, "cannot open file");
    fseek(f, 0, SEEK_END);
    const uint64_t fsize = (uint64_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    if (fsize < 8 || (fsize & 3)) {
        fclose(f);
        return fail(ROAR_ERR_FORMAT, "graph file truncated");
    }
    std::vector<uint32_t> raw((size_t)(fsize / 4));
    if (fread(raw.data(), 4, raw.size(), f) != raw.size()) {
        fclose(f);
        return fail(ROAR_ERR_FORMAT, "graph file truncated");
    }
    fclose(f);
    const uint32_t ep = raw[0], npts = raw[1];  // src/index_bipartite.cpp:2101-2104
    std::vector<uint64_t> off((size_t)npts + 1);
    std::vector<uint32_t> nb;
    nb.reserve(raw.size());
    size_t p = 2;
    for (uint32_t i = 0; i < npts; ++i) {
        if (p >= raw.size()) return fail(ROAR_ERR_FORMAT, "graph file truncated");
        uint32_t deg = raw[p++];
        if (p + deg > raw.size()) return fail(ROAR_ERR_FORMAT, "graph file truncated");
        off[i] = nb.size();
        nb.insert(nb.end(), raw.begin() + p, raw.begin() + p + deg);
        p += deg;
    }
    off[npts] = nb.size();
    static const uint32_t dummy = 0;
    return roar_set_graph_csr(idx, ep, npts, off.data(), nb.empty() ? &dummy : nb.data());
}

int roar_save_graph_file(roar_index *idx, const char *path) {
    DeviceRestore keep_device_;
    if (!idx || !path) return fail(ROAR_ERR_ARG, "bad arguments");
    if (idx->gn == 0) return fail(ROAR_ERR_ARG, "no graph loaded");
    {
        int e = use_device(idx);
        if (!e) e = ensure_host_csr(idx);
        if (e) return e;
    }
    FILE *f = fopen(path, "wb");
    if (!f) return fail(ROAR_ERR_IO, "cannot open file");  // src/index_bipartite.cpp:2608-2610
    uint32_t hdr[2] = {idx->ep, (uint32_t)idx->gn};
    fwrite(hdr, 4, 2, f);
    for (uint64_t i = 0; i < idx->gn; ++i) {
        uint32_t deg = (uint32_t)(idx->h_off[i + 1] - idx->h_off[i]);
        fwrite(&deg, 4, 1, f);
        fwrite(idx->h_nbrs.data() + idx->h_off[i], 4, deg, f);
    }
    fclose(f);
    return ROAR_OK;
}

int roar_get_graph_csr(roar_index *idx, uint32_t *ep, uint64_t *n, uint64_t *nedges, uint64_t *offsets,
                       uint32_t *nbrs) {
    DeviceRestore keep_device_;
    if (!idx) return fail(ROAR_ERR_ARG, "bad arguments");
    if (idx->gn) {
        int e = use_device(idx);
        if (!e) e = ensure_host_csr(idx);
        if (e) return e;
    }
    if (ep) *ep = idx->ep;
    if (n) *n = idx->gn;
    if (nedges) *nedges = idx->nedges;
    if (offsets && idx->gn) memcpy(offsets, idx->h_off.data(), (idx->gn + 1) * sizeof(uint64_t));
    if (nbrs && idx->nedges) memcpy(nbrs, idx->h_nbrs.data(), idx->nedges * sizeof(uint32_t));
    return ROAR_OK;
}

int roar_build_graph(roar_index *idx, const roar_build_params *bp) {
    DeviceRestore keep_device_;
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

int roar_init_visited_pool(roar_index *idx, uint32_t slots) {
    DeviceRestore keep_device_;
    if (!idx || idx->n == 0) return fail(ROAR_ERR_ARG, "InitVisitedListPool before the base vectors were loaded");
    int e = use_device(idx);
    if (e) return e;
    slots = std::max<uint32_t>(slots, 1);
    return ensure_tags(idx, &slots);
}

namespace {
int search_device(roar_index *idx, const float *queries_dev, uint64_t nq, uint32_t qstride_floats, uint32_t k, uint32_t L_pq,
                  uint32_t *ids_dev, float *dists_dev, uint32_t *cmps_dev, uint32_t *hops_dev, uint32_t *scanned_dev,
                  uint32_t *short_count_dev, void *hip_stream, const uint32_t *ready_dev, const LaunchPlan *planned = nullptr);
}

int roar_search_batch_device(roar_index *idx, const float *queries_dev, uint64_t nq, uint32_t qstride_floats,
                             uint32_t k, uint32_t L_pq, uint32_t *ids_dev, float *dists_dev, uint32_t *cmps_dev,
                             uint32_t *hops_dev, uint32_t *scanned_dev, uint32_t *short_count_dev,
                             void *hip_stream) {
    DeviceRestore keep_device_;
    return search_device(idx, queries_dev, nq, qstride_floats, k, L_pq, ids_dev, dists_dev, cmps_dev, hops_dev, scanned_dev,
                         short_count_dev, hip_stream, nullptr);
}

namespace {

// ---- launch-plan calibration -----------------------------------------------------------------------------------------------
// Which of {8-bit screen on / off} x {on-chip tag table of 2 048 ... 16 384 tags, HBM epoch tags} is fastest depends on the
// graph's degree distribution, the row length, the data (how many neighbours the screen's bound can drop), L_pq and the batch
// size.  Rounds 1-3 carried a table of thresholds measured on two synthetic graphs; now the index measures itself: the first
// search of a (L_pq band, batch band, grid cap) runs the candidates on the first queries of that very call -- results into scratch
// buffers, each candidate timed with HIP events on the caller's stream -- and keeps the winner.  Results never depend on the plan
// (ids / distances / hops are the reference's in every form), so a calibration can only cost time: 5-8 launches over at most
// 12 288 queries, once per band.  Options that fix a choice ("screen", "visited_mode", "vis_entries") take it out of the search;
// "calibrate" = 0 switches the whole thing off (plan_search's defaults).  ROAR_LOG_CALIBRATION=1 prints every decision.
// (a sample of 4 096 queries -- a third more than one residency of the chip -- misjudged the 10 000-query launch: 8 192 tags looked 5 %
// faster than 4 096 on it and were 8 % slower on the whole batch, where the table's cost is resident workgroups; T2I-10M, L_pq = 50)
constexpr uint32_t CAL_SAMPLE = 12288;  // four residencies of the chip (3 072 single-wave workgroups at twelve per CU)

bool calibratable(const roar_index *idx) {
    if (!idx->calibrate || idx->building || idx->profile) return false;
    const bool screen_free = idx->screen_mode == 0 && has_screen_kernel(idx);
    const bool mode_free = idx->visited_mode == ROAR_VISITED_AUTO;
    const bool entries_free = idx->vis_entries == 0 && (idx->visited_mode == ROAR_VISITED_AUTO || idx->visited_mode == ROAR_VISITED_LDS ||
                                                        idx->visited_mode == ROAR_VISITED_LDS_FORGETFUL);
    return screen_free || mode_free || entries_free;
}

uint64_t plan_key(const roar_index *idx, uint64_t nq, uint32_t L) {
    uint32_t lband = 0;  // half-octave bands of L_pq: [2^(b/2), 2^((b+1)/2))
    while (lband < 40 && std::pow(2.0, 0.5 * (lband + 1)) <= (double)L) lband++;
    // batch bands by queries per CU: 1, 2, 4, 6, 8, 10, 12, more -- fine around the residency limit (6-12 single-wave workgroups per CU
    // by LDS and registers), where a table's size decides whether the whole batch is resident (1 250 and 2 500 queries want different
    // tables on the benchmark graph: 0.415 / 0.431 ms with 8 192 / 4 096 tags against 0.719 / 0.559 ms)
    const uint64_t cus = (uint64_t)std::max(idx->n_cus, 1);
    const uint64_t per_cu = (nq + cus - 1) / cus;
    const uint32_t nqband = per_cu <= 1 ? 0u : per_cu <= 2 ? 1u : per_cu <= 4 ? 2u : per_cu <= 6 ? 3u : per_cu <= 8 ? 4u : per_cu <= 10 ? 5u : per_cu <= 12 ? 6u : 7u;
    // A launch with a cap on its grid (option "max_grid") is one of several that share the chip: what a table costs there is resident
    // workgroups of ALL of them together, i.e. the regime of the largest batches, whatever its own size -- it takes (and, if need be,
    // makes: uncapped) the calibration of that band.  (Timed alone under its cap, a third of the chip, every table looks the same:
    // the first build picked 16 384 tags for the capped launches of bench.py's timed loop and halved its rate.)
    const uint32_t band = idx->max_grid > 0 ? 7u : nqband;
    return (uint64_t)lband | ((uint64_t)band << 8) | ((uint64_t)(idx->waves_per_cu & 0xFF) << 20);
}

const roar_index::Calibration *find_calibration(const roar_index *idx, uint64_t key) {
    for (const auto &c : idx->plan_cache)
        if (c.key == key) return &c;
    return nullptr;
}

int launch_planned(roar_index *idx, LaunchPlan &plan, hipStream_t st);

// -> *out = the fastest candidate for launches like this one (and a cache entry); on anything that keeps the measurement from
// running (no memory for the scratch results, a candidate that cannot be planned) the defaults stand and nothing is cached.
int calibrate_plan(roar_index *idx, const float *qdev, uint64_t nq, uint32_t qstride, uint32_t k, uint32_t L, hipStream_t st,
                   PlanPick *out, bool *calibrated) {
    *calibrated = false;
    const uint64_t key = plan_key(idx, nq, L);
    const uint32_t S = (uint32_t)std::min<uint64_t>(nq, CAL_SAMPLE);
    uint32_t *sc_ids = nullptr;
    if (roar_dev_malloc((void **)&sc_ids, (size_t)S * k * 8) != hipSuccess) {
        (void)hipGetLastError();
        return ROAR_OK;
    }
    float *sc_d = reinterpret_cast<float *>(sc_ids + (size_t)S * k);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        (void)hipGetLastError();
        if (e0) (void)hipEventDestroy(e0);
        (void)hipFree(sc_ids);
        return ROAR_OK;
    }
    const bool log = getenv("ROAR_LOG_CALIBRATION") != nullptr;
    const bool had_tags = idx->tags != nullptr;
    const int cap_keep = idx->max_grid;
    idx->max_grid = 0;  // (candidates are timed on the whole chip: see plan_key)
    uint32_t runs = 0;
    int hard_error = ROAR_OK;
    // one candidate: planned for the sample, launched, timed; short launches twice (the faster counts: the first launch of a
    // kernel form also pays its code-object load).  -> ms, or a negative value if the candidate cannot run here
    auto time_pick = [&](const PlanPick &pk, uint32_t *entries_used, int *mode_used, int *screen_used) -> float {
        float best = -1.f;
        for (int rep = 0; rep < 2; ++rep) {
            LaunchPlan pl;
            const std::string keep = g_err;
            int e = plan_search(idx, S, k, L, &pl, false, &pk);
            if (e) {
                g_err = keep;
                if (e == ROAR_ERR_DEVICE) hard_error = e;
                return -1.f;
            }
            if (entries_used) *entries_used = (uint32_t)idx->last_vis_entries;
            if (mode_used) *mode_used = idx->last_mode;
            if (screen_used) *screen_used = idx->last_screen;
            pl.p.queries = qdev;
            pl.p.qstride = qstride;
            pl.p.out_ids = sc_ids;
            pl.p.out_dists = sc_d;
            if (hipEventRecord(e0, st) != hipSuccess) return -1.f;
            if ((e = launch_planned(idx, pl, st))) {
                hard_error = e;
                return -1.f;
            }
            float ms = 0.f;
            if (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) {
                hard_error = fail(ROAR_ERR_DEVICE, "calibration launch failed: %s", hipGetErrorString(hipGetLastError()));
                return -1.f;
            }
            runs++;
            best = best < 0.f ? ms : std::min(best, ms);
            if (ms > 4.0f) break;
        }
        return best;
    };
    const bool screen_free = idx->screen_mode == 0 && has_screen_kernel(idx);
    const bool mode_free = idx->visited_mode == ROAR_VISITED_AUTO;
    const bool lds_possible = idx->visited_mode != ROAR_VISITED_EXACT && (idx->n + 65533) / 65534 <= 2048;
    const bool entries_free = idx->vis_entries == 0 && lds_possible;
    struct Cand {
        PlanPick pk;
        float ms;
        uint32_t entries;
        int mode, screen;
        int protect = -1;  // (-1: whatever the default rule gave; the last stage fixes it)
    };
    std::vector<Cand> tried;
    tried.reserve(16);
    auto try_pick = [&](int screen, int mode, uint32_t entries, int protect = -1) -> const Cand * {
        PlanPick pk;
        pk.screen = screen, pk.mode = mode, pk.entries = entries, pk.protect = protect;
        Cand c{pk, -1.f, 0, 0, 0, protect};
        c.ms = time_pick(pk, &c.entries, &c.mode, &c.screen);
        if (c.ms < 0.f) return nullptr;
        for (const Cand &o : tried)  // (a table below the minimum for this many points is planned as the minimum: the same launch)
            if (o.entries == c.entries && o.mode == c.mode && o.screen == c.screen && o.protect == c.protect) return nullptr;
        if (log)
            fprintf(stderr, "[roar calibration] L_pq=%u sample=%u: screen %s, %s, %u tags%s: %.3f ms\n", L, S, c.screen ? "on" : "off",
                    c.mode == ROAR_VISITED_EXACT ? "HBM tags" : "on-chip table", c.entries,
                    c.protect == 1 ? ", pool members' tags kept" : c.protect == 0 ? ", plain replacement" : "", c.ms);
        tried.push_back(c);
        return &tried.back();
    };
    // the default plan first: what a launch without calibration would have run (the yardstick in the log)
    PlanPick none;
    float default_ms = -1.f;
    {
        uint32_t eu = 0;
        int mu = 0, su = 0;
        default_ms = time_pick(none, &eu, &mu, &su);
        if (default_ms >= 0.f) {
            PlanPick pk;
            pk.screen = su, pk.mode = mu, pk.entries = mu == ROAR_VISITED_LDS ? eu : 0;
            tried.push_back(Cand{pk, default_ms, eu, mu, su, -1});
            if (log)
                fprintf(stderr, "[roar calibration] L_pq=%u sample=%u: default plan (screen %s, %s, %u tags): %.3f ms\n", L, S,
                        su ? "on" : "off", mu == ROAR_VISITED_EXACT ? "HBM tags" : "on-chip table", eu, default_ms);
        }
    }
    if (!hard_error) {
        std::vector<int> screens;
        if (screen_free) screens = {1, 0};
        else screens = {-1};
        // stage 1: every screen choice with the on-chip table at 4 096 and 8 192 tags (or as the options fix it)
        if (lds_possible) {
            for (int sc : screens) {
                if (entries_free) {
                    try_pick(sc, ROAR_VISITED_LDS, 4096);
                    if (!hard_error) try_pick(sc, ROAR_VISITED_LDS, 8192);
                } else {
                    try_pick(sc, mode_free ? ROAR_VISITED_LDS : -1, 0);
                }
                if (hard_error) break;
            }
        }
        auto best_of = [&]() -> const Cand * {
            const Cand *b = nullptr;
            for (const Cand &c : tried)
                if (!b || c.ms < b->ms) b = &c;
            return b;
        };
        // stage 2: further in the direction that won, while it keeps winning (a smaller table buys resident workgroups -- a long pool
        // leaves room for nine to twelve per CU and every 1 280 bytes count -- a bigger one forgets less), then the 3 x 2^k size next to
        // the winner
        if (!hard_error && entries_free) {
            const Cand *b = best_of();
            if (b && b->mode == ROAR_VISITED_LDS) {
                const int sc = screen_free ? b->screen : -1;
                if (b->entries <= 4096) {
                    for (uint32_t e = 2048; e >= 1024 && !hard_error; e >>= 1) {
                        const Cand *c = try_pick(sc, ROAR_VISITED_LDS, e);
                        b = best_of();
                        if (!c || !b || b->entries != c->entries || b->mode != ROAR_VISITED_LDS) break;  // (did not win: stop here)
                    }
                } else if (b->entries >= 8192) {
                    try_pick(sc, ROAR_VISITED_LDS, 16384);
                }
                b = best_of();
                if (!hard_error && b && b->mode == ROAR_VISITED_LDS && L >= 256 && b->entries >= 1024 && b->entries <= 4096) {
                    const uint32_t e0 = b->entries;
                    try_pick(sc, ROAR_VISITED_LDS, e0 + e0 / 2);
                    if (!hard_error && (e0 & (e0 - 1)) == 0) try_pick(sc, ROAR_VISITED_LDS, e0 - e0 / 4);
                }
            }
        }
        // stage 3: HBM epoch tags instead of the table -- only worth a look (and the tag lists' memory) once the pool is long
        if (!hard_error && mode_free && L >= 512) {
            const Cand *b = best_of();
            try_pick(screen_free ? (b ? b->screen : 1) : -1, ROAR_VISITED_EXACT, 0);
        }
        if (!lds_possible && screen_free && !hard_error)
            for (int sc : screens) try_pick(sc, -1, 0);
        // last stage: the winner with the other setting of "tag_protect" (only an on-chip table under a pool in LDS has one)
        if (!hard_error && idx->tag_protect < 0 && L > 64) {
            const Cand *b = best_of();
            if (b && b->mode == ROAR_VISITED_LDS) {
                PlanPick pk = b->pk;  // (planned once more, untimed, to learn which setting the winner ran with)
                LaunchPlan pl;
                const std::string keep = g_err;
                if (plan_search(idx, S, k, L, &pl, false, &pk) == ROAR_OK) try_pick(pk.screen, pk.mode, pk.entries, idx->last_tag_protect ? 0 : 1);
                else g_err = keep;
            }
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(sc_ids);
    idx->max_grid = cap_keep;
    if (hard_error) return hard_error;
    const Cand *best = nullptr;
    for (const Cand &c : tried)
        if (!best || c.ms < best->ms) best = &c;
    // the HBM tag lists (up to half of the free memory) were made for a candidate: unless it won, or they were there before, they go
    if (!had_tags && idx->tags && !(best && best->mode == ROAR_VISITED_EXACT)) {
        (void)hipFree(idx->tags);
        (void)hipFree(idx->epochs);
        idx->tags = nullptr;
        idx->epochs = nullptr;
        idx->slots = 0;
        idx->last_used_tags = false;
    }
    if (!best) return ROAR_OK;
    // Among the candidates within 1.5 % of the fastest (noise), the one that counts best wins: HBM tags return the reference's cmps
    // exactly, a bigger on-chip table forgets less (cmps >= the reference's; ids / distances / hops never depend on any of this) --
    // and, all else equal, the default plan.
    {
        const float lim = best->ms * 1.015f;
        auto fidelity = [&](const Cand &c) -> uint64_t {
            return ((uint64_t)(c.mode == ROAR_VISITED_EXACT ? 1 : 0) << 40) | ((uint64_t)c.entries << 8) | (uint64_t)(&c == &tried.front() ? 1 : 0);
        };
        for (const Cand &c : tried)
            if (c.ms <= lim && c.screen == best->screen && fidelity(c) > fidelity(*best)) best = &c;
    }
    out->screen = screen_free ? best->screen : -1;
    out->mode = mode_free ? best->mode : -1;
    out->entries = (entries_free && best->mode == ROAR_VISITED_LDS) ? best->entries : 0;
    out->protect = idx->tag_protect < 0 ? best->protect : -1;
    *calibrated = true;
    try {
        roar_index::Calibration c{key, out->screen, out->mode, out->entries, out->protect, best->ms, default_ms, S, L, runs};
        idx->plan_cache.push_back(c);
    } catch (...) {
    }
    if (log)
        fprintf(stderr, "[roar calibration] L_pq=%u nq=%llu key=%llx -> screen %s, %s, %u tags%s: %.3f ms (default plan %.3f ms), %u launches\n", L,
                (unsigned long long)nq, (unsigned long long)key, best->screen ? "on" : "off",
                best->mode == ROAR_VISITED_EXACT ? "HBM tags" : "on-chip table", best->entries,
                best->protect == 1 ? ", pool members' tags kept" : best->protect == 0 ? ", plain replacement" : "", best->ms, default_ms, runs);
    return ROAR_OK;
}

// The plan of a launch: calibrated choices where there are any (measuring them now if `qdev` is given and this kind of launch
// has not been seen yet), defaults otherwise.  *need_queries (may be null): set when a calibration is due but qdev is null.
int resolve_plan(roar_index *idx, const float *qdev, uint64_t nq, uint32_t qstride, uint32_t k, uint32_t L, hipStream_t st, LaunchPlan *plan,
                 bool *need_queries) {
    if (need_queries) *need_queries = false;
    idx->last_calibrated = 0;
    if (!calibratable(idx)) return plan_search(idx, nq, k, L, plan);
    // (argument errors come from plan_search, not from a calibration run)
    if (!idx->base || !idx->adj || idx->gn != idx->n || L == 0 || k == 0 || k > L) return plan_search(idx, nq, k, L, plan);
    PlanPick pick;
    const uint64_t key = plan_key(idx, nq, L);
    if (const roar_index::Calibration *c = find_calibration(idx, key)) {
        pick.screen = c->screen, pick.mode = c->mode, pick.entries = c->entries, pick.protect = c->protect;
        idx->last_calibrated = 1;
    } else if (qdev) {
        bool done = false;
        int e = calibrate_plan(idx, qdev, nq, qstride, k, L, st, &pick, &done);
        if (e) return e;
        idx->last_calibrated = done ? 1 : 0;
    } else if (need_queries) {
        *need_queries = true;
        return ROAR_OK;
    }
    return plan_search(idx, nq, k, L, plan, false, &pick);
}
}  // namespace

namespace {
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
    plan.p.counter = idx->ring + 4 * (size_t)(idx->launch_seq.fetch_add(1) % LAUNCH_RING);
    int rc = roar_launch_search(&plan.p, &plan.sel, plan.grid, st);
    if (rc != 0) return fail(ROAR_ERR_DEVICE, "search kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return ROAR_OK;
}

int search_host(roar_index *idx, const float *queries, uint64_t nq, uint32_t qstride_floats, uint32_t k, uint32_t L_pq,
                uint32_t *ids, float *dists, uint32_t *cmps, uint32_t *hops, uint32_t *scanned);

// The reference's harness calls SearchRoarGraph once per query from num_threads OpenMP threads
// (tests/test_search_roargraph.cpp:203-209).  One-query launches would leave the GPU idle, so concurrent
// one-query calls are combined: a caller that finds no leader becomes the leader, takes every request queued with
// the same (k, L_pq) as the oldest one -- those that arrived while the previous launch was in flight -- runs them as
// one batch and hands each caller its own rows.  A leader serves exactly one batch: when its own request is done it
// returns, and a caller still waiting takes over (so no caller's latency grows with the load of the others).  Nothing
// thrown inside a leader's turn escapes the C ABI or leaves the queue without a leader: the batch is failed instead.
// Results are those of the batch path, i.e. of the reference, query by query.
int search_one(roar_index *idx, roar_index::OneQuery &req) {
    std::unique_lock<std::mutex> lk(idx->cq_mu);
    try {
        idx->cq.push_back(&req);
    } catch (...) {
        return fail(ROAR_ERR_DEVICE, "out of host memory");
    }
    while (!req.done) {
        if (idx->cq_leader) {
            idx->cq_cv.wait(lk, [&] { return req.done || !idx->cq_leader; });
            continue;
        }
        idx->cq_leader = true;
        std::vector<roar_index::OneQuery *> batch;
        int rc = ROAR_OK;
        std::string err;
        try {
            std::vector<roar_index::OneQuery *> rest;
            const roar_index::OneQuery *head = idx->cq.front();
            for (roar_index::OneQuery *r : idx->cq)
                ((r->k == head->k && r->L == head->L && batch.size() < 4096) ? batch : rest).push_back(r);
            idx->cq.swap(rest);
            lk.unlock();
            const uint64_t m = batch.size();
            const uint32_t k = head->k, dp = idx->dim_pad;
            std::vector<float> qbuf((size_t)m * dp, 0.0f);
            for (uint64_t i = 0; i < m; ++i)
                memcpy(&qbuf[i * dp], batch[i]->q, std::min(batch[i]->qstride, idx->dim) * sizeof(float));
            std::vector<uint32_t> r_ids((size_t)m * k), r_cmps(m), r_hops(m), r_scan(m);
            std::vector<float> r_d((size_t)m * k);
            rc = search_host(idx, qbuf.data(), m, dp, k, head->L, r_ids.data(), r_d.data(), r_cmps.data(), r_hops.data(),
                             r_scan.data());
            err = g_err;
            for (uint64_t i = 0; i < m; ++i) {
                roar_index::OneQuery *r = batch[i];
                r->rc = rc;
                if (rc == ROAR_OK || rc == ROAR_ERR_SHORT) {
                    memcpy(r->ids, &r_ids[i * k], k * sizeof(uint32_t));
                    memcpy(r->dists, &r_d[i * k], k * sizeof(float));
                    if (r->cmps) *r->cmps = r_cmps[i];
                    if (r->hops) *r->hops = r_hops[i];
                    if (r->scanned) *r->scanned = r_scan[i];
                    uint32_t have = 0;
                    while (have < k && r_ids[i * k + have] != 0xFFFFFFFFu) have++;
                    r->rc = ROAR_OK;
                    if (have < k) {  // this query's own shortfall, worded as the reference throws it (:2408-2412)
                        char msg[96];
                        snprintf(msg, sizeof msg, "not enough results: %u, expected: %u", have, k);
                        r->rc = ROAR_ERR_SHORT;
                        r->err = msg;
                    }
                } else {
                    r->err = err;
                }
            }
        } catch (...) {  // std::bad_alloc in the staging vectors: fail this batch, keep the queue alive
            for (roar_index::OneQuery *r : batch) {
                r->rc = ROAR_ERR_DEVICE;
                try {
                    r->err = "out of host memory while combining one-query calls";
                } catch (...) {
                }
            }
        }
        if (!lk.owns_lock()) lk.lock();
        idx->last_combined = (int64_t)batch.size();
        for (roar_index::OneQuery *r : batch) r->done = true;
        idx->cq_leader = false;
        idx->cq_cv.notify_all();
    }
    if (req.rc != ROAR_OK) g_err = req.err;
    return req.rc;
}
}  // namespace

int roar_search_batch(roar_index *idx, const float *queries, uint64_t nq, uint32_t qstride_floats, uint32_t k,
                      uint32_t L_pq, uint32_t *ids, float *dists, uint32_t *cmps, uint32_t *hops,
                      uint32_t *scanned) {
    DeviceRestore keep_device_;
    if (!idx || !ids || !dists || (nq && !queries)) return fail(ROAR_ERR_ARG, "bad search arguments");
    if (nq == 0) return ROAR_OK;
    if (qstride_floats < idx->dim) return fail(ROAR_ERR_ARG, "query stride smaller than the dimension");
    if (nq == 1 && idx->combine && k > 0 && k <= 4096) {
        roar_index::OneQuery req;
        req.q = queries, req.qstride = qstride_floats, req.k = k, req.L = L_pq;
        req.ids = ids, req.dists = dists, req.cmps = cmps, req.hops = hops, req.scanned = scanned;
        return search_one(idx, req);
    }
    return search_host(idx, queries, nq, qstride_floats, k, L_pq, ids, dists, cmps, hops, scanned);
}

namespace {
int search_host(roar_index *idx, const float *queries, uint64_t nq, uint32_t qstride_floats, uint32_t k, uint32_t L_pq,
                uint32_t *ids, float *dists, uint32_t *cmps, uint32_t *hops, uint32_t *scanned) {
    std::lock_guard<std::mutex> guard(idx->mu);
    int e = use_device(idx);
    if (e) return e;
    const float *qdev = nullptr;
    uint32_t qs = 0;
    bool deferred = false;
    if ((e = stage_queries(idx, queries, nq, qstride_floats, k, &qdev, &qs, &deferred))) return e;
    LaunchPlan probe;
    bool have_plan = false;
    if (deferred) {
        // the kernel forms that can wait for the packing thread: one wave per query, no hub bitmap, not instrumented
        // (the plan made here is the one the launch uses: planning can be expensive -- tag pools, hub rows)
        bool need_queries = false;
        if ((e = resolve_plan(idx, nullptr, nq, qs, k, L_pq, idx->stream, &probe, &need_queries))) return e;
        if (need_queries) {  // a launch plan calibration is due: it reads the queries, so they are packed in front of it
            pack_queries(idx, queries, nq, qstride_floats);
            deferred = false;
        } else {
            have_plan = true;
            const bool gate = probe.sel.form == ROAR_FORM_SEARCH && probe.sel.waves == 1 && !probe.sel.hub && !probe.sel.prof &&
                              !(idx->traffic_dev && nq <= idx->traffic_cap);
            if (!gate) {  // pack in front of the launch
                pack_queries(idx, queries, nq, qstride_floats);
                deferred = false;
            }
        }
    }
    uint32_t *d_ids = idx->out_dev;
    float *d_dists = reinterpret_cast<float *>(d_ids + nq * k);
    uint32_t *d_cmps = d_ids + 2 * nq * k;
    uint32_t *d_hops = d_cmps + nq;
    uint32_t *d_scan = d_hops + nq;
    uint32_t *d_short = d_scan + nq;
    // The gated launch waits for rows that are packed behind it -- by a helper thread that is started BEFORE the launch,
    // so that a launch call that only returns when the kernel has finished (HIP_LAUNCH_BLOCKING, AMD_SERIALIZE_KERNEL,
    // a debugger or a tool that serialises dispatches) still finds its rows arriving.  No helper thread, no gate.
    std::thread packer;
    if (deferred) {
        try {
            packer = std::thread(pack_queries, idx, queries, nq, qstride_floats);
        } catch (...) {
            pack_queries(idx, queries, nq, qstride_floats);
            deferred = false;
        }
    }
    e = search_device(idx, qdev, nq, qs, k, L_pq, d_ids, d_dists, d_cmps, d_hops, d_scan, d_short, idx->stream,
                      deferred ? idx->ready_dev : nullptr, have_plan ? &probe : nullptr);
    if (packer.joinable()) packer.join();  // (plain memcpy: it always runs to the end, launch or no launch)
    if (e) return e;
    return fetch_results(idx, nq, k, ids, dists, cmps, hops, scanned);
}
}  // namespace

// ---- replicated index, one query block split over g devices (BASELINE configs[3]) --------------------

int roar_clone_to_device(roar_index *src, int device, roar_index **out) {
    DeviceRestore keep_device_;
    if (!src || !out) return fail(ROAR_ERR_ARG, "bad arguments");
    *out = nullptr;
    if (!src->base || src->n == 0) return fail(ROAR_ERR_ARG, "clone before base vectors were loaded");
    roar_index *dst = nullptr;
    int e = roar_create(&dst, src->dim, src->metric, device);
    if (e) return e;
    auto bail = [&](int code) {
        const std::string keep = g_err;
        roar_destroy(dst);
        g_err = keep;
        return code;
    };
    std::lock_guard<std::mutex> guard(src->mu);
    dst->visited_mode = src->visited_mode;
    dst->waves_per_cu = src->waves_per_cu;
    dst->vis_entries = src->vis_entries;
    dst->max_grid = src->max_grid;
    dst->search_kernel = src->search_kernel;
    dst->pool_form = src->pool_form;
    dst->combine = src->combine;
    dst->hub_mode = src->hub_mode;
    dst->profile = src->profile;
    dst->screen_mode = src->screen_mode;
    dst->calibrate = src->calibrate;
    if ((e = alloc_base(dst, src->n))) return bail(e);
    // rows are re-strided to the padded dimension if the source adopted a wider block
    hipError_t he = hipSuccess;
    if (src->ld == dst->ld) {
        he = hipMemcpyPeer(dst->base, dst->device, src->base, src->device, (size_t)src->n * src->ld * sizeof(float));
    } else {
        he = hipMemcpy2D(dst->base, (size_t)dst->ld * 4, src->base, (size_t)src->ld * 4, (size_t)dst->ld * 4, (size_t)src->n,
                         hipMemcpyDefault);
    }
    if (he != hipSuccess) return bail(fail(ROAR_ERR_DEVICE, "copy of the base rows to device %d failed: %s", device, hipGetErrorString(he)));
    if (src->adj) {
        const size_t words = (size_t)src->gn * src->adj_stride + ADJ_SLACK_WORDS;
        if (roar_dev_malloc((void **)&dst->adj, words * sizeof(uint32_t)) != hipSuccess)
            return bail(fail(ROAR_ERR_DEVICE, "out of device memory for the graph replica"));
        he = hipMemcpyPeer(dst->adj, dst->device, src->adj, src->device, words * sizeof(uint32_t));
        if (he != hipSuccess) return bail(fail(ROAR_ERR_DEVICE, "copy of the graph to device %d failed: %s", device, hipGetErrorString(he)));
        dst->adj_stride = src->adj_stride;
        dst->ep = src->ep;
        dst->max_deg = src->max_deg;
        dst->gn = src->gn;
        dst->nedges = src->nedges;  // the host CSR copy is rebuilt from the device rows if it is ever asked for
    }
    he = hipDeviceSynchronize();
    if (he != hipSuccess) return bail(fail(ROAR_ERR_DEVICE, "replica on device %d: %s", device, hipGetErrorString(he)));
    *out = dst;
    return ROAR_OK;
}

int roar_search_batch_multi(roar_index *const *replicas, uint32_t g, const float *queries, uint64_t nq, uint32_t qstride_floats,
                            uint32_t k, uint32_t L_pq, uint32_t *ids, float *dists, uint32_t *cmps, uint32_t *hops,
                            uint32_t *scanned) {
    DeviceRestore keep_device_;
    if (!replicas || g == 0 || g > 64) return fail(ROAR_ERR_ARG, "between 1 and 64 replicas");
    if (!ids || !dists || (nq && !queries)) return fail(ROAR_ERR_ARG, "bad search arguments");
    for (uint32_t r = 0; r < g; ++r) {
        if (!replicas[r]) return fail(ROAR_ERR_ARG, "replica %u is null", r);
        if (replicas[r]->dim != replicas[0]->dim || replicas[r]->n != replicas[0]->n || replicas[r]->gn != replicas[0]->gn ||
            replicas[r]->metric != replicas[0]->metric)
            return fail(ROAR_ERR_ARG, "replica %u does not hold the same index as replica 0", r);
    }
    if (nq == 0) return ROAR_OK;
    if (qstride_floats < replicas[0]->dim) return fail(ROAR_ERR_ARG, "query stride smaller than the dimension");
    // replica r takes queries [r * nq / g, (r + 1) * nq / g) and writes the same rows of the caller's arrays
    std::vector<int> rc(g, ROAR_OK);
    std::vector<std::string> msg(g);
    auto work = [&](uint32_t r) {
        const uint64_t lo = nq * r / g, hi = nq * (r + 1) / g;
        if (hi == lo) return;
        try {
            rc[r] = search_host(replicas[r], queries + lo * qstride_floats, hi - lo, qstride_floats, k, L_pq, ids + lo * k,
                                dists + lo * k, cmps ? cmps + lo : nullptr, hops ? hops + lo : nullptr,
                                scanned ? scanned + lo : nullptr);
            if (rc[r] != ROAR_OK) msg[r] = g_err;
        } catch (...) {
            rc[r] = ROAR_ERR_DEVICE;
            msg[r] = "out of host memory in a replica's search";
        }
    };
    {
        std::vector<std::thread> th;
        try {
            for (uint32_t r = 1; r < g; ++r) th.emplace_back(work, r);
        } catch (...) {
            for (auto &x : th) x.join();
            return fail(ROAR_ERR_DEVICE, "cannot start the per-device host threads");
        }
        work(0);
        for (auto &x : th) x.join();
    }
    int first_short = -1;
    for (uint32_t r = 0; r < g; ++r) {
        if (rc[r] == ROAR_ERR_SHORT) {
            if (first_short < 0) first_short = (int)r;
        } else if (rc[r] != ROAR_OK) {
            g_err = msg[r];
            return rc[r];
        }
    }
    if (first_short >= 0) {
        g_err = msg[first_short];
        return ROAR_ERR_SHORT;
    }
    return ROAR_OK;
}

// ---- pinned host memory for query / result blocks ----------------------------------------------------------------------

void *roar_host_alloc(uint64_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, (size_t)(bytes ? bytes : 1), hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        fail(ROAR_ERR_DEVICE, "pinned host allocation of %llu bytes failed", (unsigned long long)bytes);
        return nullptr;
    }
    return p;
}

void roar_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int roar_host_register(void *p, uint64_t bytes) {
    if (!p || bytes == 0) return fail(ROAR_ERR_ARG, "bad arguments");
    const hipError_t e = hipHostRegister(p, (size_t)bytes, hipHostRegisterMapped | hipHostRegisterPortable);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(ROAR_ERR_DEVICE, "hipHostRegister of %llu bytes failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
    }
    return ROAR_OK;
}

int roar_host_unregister(void *p) {
    if (!p) return fail(ROAR_ERR_ARG, "bad arguments");
    const hipError_t e = hipHostUnregister(p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(ROAR_ERR_DEVICE, "hipHostUnregister failed: %s", hipGetErrorString(e));
    }
    return ROAR_OK;
}


// ---- bipartite two-hop variant -------------------------------------------------

namespace {
// Every row of a base node must hold training-query nodes and every row of a training-query node base
// nodes: SearchBipartiteGraph scores graph[graph[cur]] entries as base rows (index_bipartite.cpp:330-340),
// anything else reads outside data_bp_ in the reference.
int check_bipartite(roar_index *idx) {
    if (idx->bp_checked) return ROAR_OK;
    const uint64_t nb = idx->n, nt = idx->bp_n;
    if (nt < nb)
        return fail(ROAR_ERR_FORMAT, "bipartite graph has %llu nodes, fewer than the %llu base rows",
                    (unsigned long long)nt, (unsigned long long)nb);
    for (uint64_t v = 0; v < nt; ++v)
        for (uint64_t j = idx->hb_off[v]; j < idx->hb_off[v + 1]; ++j) {
            const uint32_t t = idx->hb_nbrs[j];
            if (v < nb ? (t < nb || t >= nt) : (t >= nb))
                return fail(ROAR_ERR_FORMAT, "bipartite graph: node %llu links to %u on its own side",
                            (unsigned long long)v, t);
        }
    idx->bp_checked = true;
    return ROAR_OK;
}

inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
}  // namespace

int roar_set_bipartite_csr(roar_index *idx, uint64_t n_total, const uint64_t *offsets, const uint32_t *nbrs) {
    DeviceRestore keep_device_;
    if (!idx || !offsets || (n_total && !nbrs && offsets[n_total] != 0)) return fail(ROAR_ERR_ARG, "bad graph arguments");
    if (n_total == 0 || n_total >= (1ull << 32)) return fail(ROAR_ERR_ARG, "graph must have between 1 and 2^32 - 1 nodes");
    for (uint64_t i = 0; i < n_total; ++i)
        if (offsets[i + 1] < offsets[i]) return fail(ROAR_ERR_FORMAT, "CSR offsets are not monotone");
    const uint64_t ne = offsets[n_total];
    for (uint64_t j = 0; j < ne; ++j)
        if (nbrs[j] >= n_total) return fail(ROAR_ERR_FORMAT, "neighbour id %u outside the graph", nbrs[j]);
    std::lock_guard<std::mutex> guard(idx->mu);
    int e = use_device(idx);
    if (e) return e;
    free_bipartite(idx);
    idx->hb_off.assign(offsets, offsets + n_total + 1);
    idx->hb_nbrs.assign(nbrs, nbrs + ne);
    idx->bp_n = n_total;
    idx->bp_edges = ne;
    if (idx->n && (e = check_bipartite(idx))) {
        free_bipartite(idx);
        return e;
    }
    HIP_TRY(roar_dev_malloc((void **)&idx->bp_off, (size_t)(n_total + 1) * sizeof(uint64_t)));
    HIP_TRY(roar_dev_malloc((void **)&idx->bp_nbrs, (size_t)(ne + 64) * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(idx->bp_off, offsets, (size_t)(n_total + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    if (ne) HIP_TRY(hipMemcpy(idx->bp_nbrs, nbrs, (size_t)ne * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(idx->bp_nbrs + ne, 0, 64 * sizeof(uint32_t)));
    return ROAR_OK;
}

int roar_load_bipartite_file(roar_index *idx, const char *path) {
    if (!idx || !path) return fail(ROAR_ERR_ARG, "bad arguments");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(ROAR_ERR_IO, "cannot open file");
    fseek(f, 0, SEEK_END);
    const uint64_t fsize = (uint64_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    if (fsize < 4 || (fsize & 3)) {
        fclose(f);
        return fail(ROAR_ERR_FORMAT, "graph file truncated");
    }
    std::vector<uint32_t> raw((size_t)(fsize / 4));
    if (fread(raw.data(), 4, raw.size(), f) != raw.size()) {
        fclose(f);
        return fail(ROAR_ERR_FORMAT, "graph file truncated");
    }
    fclose(f);
    const uint32_t npts = raw[0];  // src/index_bipartite.cpp:2065-2067: no entry point word in this format
    std::vector<uint64_t> off((size_t)npts + 1);
    std::vector<uint32_t> nb;
    nb.reserve(raw.size());
    size_t p = 1;
    for (uint32_t i = 0; i < npts; ++i) {
        if (p >= raw.size()) return fail(ROAR_ERR_FORMAT, "graph file truncated");
        uint32_t deg = raw[p++];
        if (p + deg > raw.size()) return fail(ROAR_ERR_FORMAT, "graph file truncated");
        off[i] = nb.size();
        nb.insert(nb.end(), raw.begin() + p, raw.begin() + p + deg);
        p += deg;
    }
    off[npts] = nb.size();
    static const uint32_t dummy = 0;
    return roar_set_bipartite_csr(idx, npts, off.data(), nb.empty() ? &dummy : nb.data());
}

int roar_save_bipartite_file(roar_index *idx, const char *path) {
    if (!idx || !path) return fail(ROAR_ERR_ARG, "bad arguments");
    if (idx->bp_n == 0) return fail(ROAR_ERR_ARG, "no bipartite graph loaded");
    FILE *f = fopen(path, "wb");
    if (!f) return fail(ROAR_ERR_IO, "cannot open file");
    uint32_t npts = (uint32_t)idx->bp_n;  // src/index_bipartite.cpp:2049-2051
    fwrite(&npts, 4, 1, f);
    for (uint64_t i = 0; i < idx->bp_n; ++i) {
        uint32_t deg = (uint32_t)(idx->hb_off[i + 1] - idx->hb_off[i]);
        fwrite(&deg, 4, 1, f);
        fwrite(idx->hb_nbrs.data() + idx->hb_off[i], 4, deg, f);
    }
    fclose(f);
    return ROAR_OK;
}

int roar_get_bipartite_csr(roar_index *idx, uint64_t *n_total, uint64_t *nedges, uint64_t *offsets, uint32_t *nbrs) {
    if (!idx) return fail(ROAR_ERR_ARG, "bad arguments");
    if (idx->bp_n == 0) return fail(ROAR_ERR_ARG, "no bipartite graph loaded");
    if (n_total) *n_total = idx->bp_n;
    if (nedges) *nedges = idx->bp_edges;
    if (offsets) memcpy(offsets, idx->hb_off.data(), (size_t)(idx->bp_n + 1) * sizeof(uint64_t));
    if (nbrs && idx->bp_edges) memcpy(nbrs, idx->hb_nbrs.data(), (size_t)idx->bp_edges * sizeof(uint32_t));
    return ROAR_OK;
}

// qbaseNNbipartite (src/index_bipartite.cpp:235-280): training query sq (node n + sq) keeps its first
// M_pjbp nearest base points; the nearest is its target, the others its row; the target's row gains
// n + sq.  The reference appends to the target rows under per-node locks from an OpenMP loop, so their
// order is run-dependent there; here it is ascending sq.  (Within a hop the search is order-free.)
int roar_build_bipartite(roar_index *idx, const uint32_t *learn_knn, uint64_t n_learn, uint32_t K, uint32_t M_pjbp) {
    if (!idx || !learn_knn) return fail(ROAR_ERR_ARG, "bad arguments");
    if (!idx->base || idx->n == 0) return fail(ROAR_ERR_ARG, "build before base vectors were loaded");
    if (n_learn == 0 || K == 0 || M_pjbp == 0) return fail(ROAR_ERR_ARG, "n_learn, K and M_pjbp must be positive");
    const uint64_t nb = idx->n, nt = nb + n_learn;
    if (nt >= (1ull << 32)) return fail(ROAR_ERR_ARG, "base + training queries exceed 2^32 - 1 nodes");
    const uint32_t keep = std::min(K, M_pjbp);  // :256-259
    std::vector<uint64_t> off((size_t)nt + 1, 0);
    for (uint64_t sq = 0; sq < n_learn; ++sq) {
        const uint32_t *nn = learn_knn + sq * K;
        const uint32_t tgt = nn[0];
        for (uint32_t i = 0; i < keep; ++i)
            if (nn[i] >= nb)
                return fail(ROAR_ERR_FORMAT, "kNN list %llu names base id %u of %llu", (unsigned long long)sq, nn[i],
                            (unsigned long long)nb);
        uint32_t len = 0;
        for (uint32_t i = 0; i < keep; ++i) len += nn[i] != tgt;  // :264-268
        off[nb + sq + 1] = len;
        off[tgt + 1] += 1;
    }
    for (uint64_t v = 0; v < nt; ++v) off[v + 1] += off[v];
    std::vector<uint32_t> nbrs((size_t)off[nt] + 1);
    std::vector<uint64_t> fill(off.begin(), off.begin() + nb);
    for (uint64_t sq = 0; sq < n_learn; ++sq) {
        const uint32_t *nn = learn_knn + sq * K;
        const uint32_t tgt = nn[0];
        uint64_t w = off[nb + sq];
        for (uint32_t i = 0; i < keep; ++i)
            if (nn[i] != tgt) nbrs[w++] = nn[i];
        nbrs[fill[tgt]++] = (uint32_t)(nb + sq);  // :269-272
    }
    return roar_set_bipartite_csr(idx, nt, off.data(), nbrs.data());
}

// Entry points for SearchBipartiteGraph: the reference draws 10 uniform base ids per query from
// std::random_device (src/index_bipartite.cpp:287-294); a counter-based generator takes its place so
// that a search can be replayed against the CPU checker.  out[q * n_init + i].
int roar_bipartite_entry_ids(uint64_t seed, uint64_t nq, uint32_t n_init, uint64_t n_base, uint32_t *out) {
    if (!out || n_base == 0 || n_base >= (1ull << 32)) return fail(ROAR_ERR_ARG, "bad arguments");
    for (uint64_t q = 0; q < nq; ++q)
        for (uint32_t i = 0; i < n_init; ++i)
            out[q * n_init + i] = (uint32_t)(splitmix64(seed ^ splitmix64(q * 64 + i)) % n_base);
    return ROAR_OK;
}

int roar_search_bipartite_batch_device(roar_index *idx, const float *queries_dev, uint64_t nq, uint32_t qstride_floats,
                                       uint32_t k, uint32_t L_pq, const uint32_t *init_ids_dev, uint32_t n_init,
                                       uint32_t *ids_dev, float *dists_dev, uint32_t *cmps_dev, uint32_t *hops_dev,
                                       uint32_t *scanned_dev, uint32_t *short_count_dev, void *hip_stream) {
    DeviceRestore keep_device_;
    if (!idx || !queries_dev || !ids_dev || !dists_dev || !init_ids_dev) return fail(ROAR_ERR_ARG, "bad search arguments");
    if (nq == 0) return ROAR_OK;
    if (n_init == 0 || n_init > 64) return fail(ROAR_ERR_ARG, "between 1 and 64 entry points per query");
    if (qstride_floats < idx->dim_pad || (qstride_floats & 3))
        return fail(ROAR_ERR_ARG, "device queries must be padded to %u floats (stride multiple of 4)", idx->dim_pad);
    int e = use_device(idx);
    if (e) return e;
    if (idx->bp_n && idx->n && (e = check_bipartite(idx))) return e;
    LaunchPlan plan;
    if ((e = plan_search(idx, nq, k, L_pq, &plan, true))) return e;
    RoarSearchParams &p = plan.p;
    p.queries = queries_dev;
    p.qstride = qstride_floats;
    p.init_ids = init_ids_dev;
    p.n_init = n_init;
    p.out_ids = ids_dev;
    p.out_dists = dists_dev;
    p.out_cmps = cmps_dev;
    p.out_hops = hops_dev;
    p.out_scanned = scanned_dev;
    p.short_count = short_count_dev;
    hipStream_t st = (hipStream_t)hip_stream;
    HIP_TRY(hipMemsetAsync(idx->counters, 0, 2 * sizeof(uint32_t), st));
    if (short_count_dev) HIP_TRY(hipMemsetAsync(short_count_dev, 0, sizeof(uint32_t), st));
    HIP_TRY(hipEventRecord(idx->ev0, st));
    int rc = roar_launch_search(&p, &plan.sel, plan.grid, st);
    if (rc != 0) return fail(ROAR_ERR_DEVICE, "search kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    HIP_TRY(hipEventRecord(idx->ev1, st));
    idx->ev_valid = true;
    return ROAR_OK;
}

int roar_search_bipartite_batch(roar_index *idx, const float *queries, uint64_t nq, uint32_t qstride_floats, uint32_t k,
                                uint32_t L_pq, const uint32_t *init_ids, uint32_t n_init, uint32_t *ids, float *dists,
                                uint32_t *cmps, uint32_t *hops) {
    DeviceRestore keep_device_;
    if (!idx || !ids || !dists || (nq && !queries)) return fail(ROAR_ERR_ARG, "bad search arguments");
    if (nq == 0) return ROAR_OK;
    if (qstride_floats < idx->dim) return fail(ROAR_ERR_ARG, "query stride smaller than the dimension");
    if (n_init == 0 || n_init > 64) return fail(ROAR_ERR_ARG, "between 1 and 64 entry points per query");
    if (!idx->base || idx->n == 0) return fail(ROAR_ERR_ARG, "search before base vectors were loaded");
    std::lock_guard<std::mutex> guard(idx->mu);
    int e = use_device(idx);
    if (e) return e;
    std::vector<uint32_t> gen;
    if (!init_ids) {  // std::random_device in the reference (:287-294): a fresh draw per call
        gen.resize((size_t)nq * n_init);
        if ((e = roar_bipartite_entry_ids(idx->bp_seed, nq, n_init, idx->n, gen.data()))) return e;
        idx->bp_seed = splitmix64(idx->bp_seed);
        init_ids = gen.data();
    } else {
        for (uint64_t i = 0; i < nq * n_init; ++i)
            if (init_ids[i] >= idx->n) return fail(ROAR_ERR_ARG, "entry point %u outside the base set", init_ids[i]);
    }
    if (idx->init_cap < nq * n_init) {
        if (idx->init_dev) (void)hipFree(idx->init_dev);
        idx->init_dev = nullptr;
        HIP_TRY(roar_dev_malloc((void **)&idx->init_dev, (size_t)nq * n_init * sizeof(uint32_t)));
        idx->init_cap = nq * n_init;
    }
    HIP_TRY(hipMemcpyAsync(idx->init_dev, init_ids, (size_t)nq * n_init * sizeof(uint32_t), hipMemcpyHostToDevice, idx->stream));
    HIP_TRY(hipStreamSynchronize(idx->stream));  // init_ids may be a temporary of this call
    const float *qdev = nullptr;
    uint32_t qs = 0;
    bool deferred = false;
    if ((e = stage_queries(idx, queries, nq, qstride_floats, k, &qdev, &qs, &deferred))) return e;
    if (deferred) pack_queries(idx, queries, nq, qstride_floats);  // (the two-hop kernels do not wait for a packing thread)
    uint32_t *d_ids = idx->out_dev;
    float *d_dists = reinterpret_cast<float *>(d_ids + nq * k);
    uint32_t *d_cmps = d_ids + 2 * nq * k;
    uint32_t *d_hops = d_cmps + nq;
    uint32_t *d_scan = d_hops + nq;
    uint32_t *d_short = d_scan + nq;
    e = roar_search_bipartite_batch_device(idx, qdev, nq, qs, k, L_pq, idx->init_dev, n_init, d_ids, d_dists, d_cmps,
                                           d_hops, d_scan, d_short, idx->stream);
    if (e) return e;
    return fetch_results(idx, nq, k, ids, dists, cmps, hops, nullptr);
}

int roar_distances(roar_index *idx, const float *queries, uint64_t nq, uint32_t qstride_floats, const uint32_t *ids,
                   uint32_t m, float *out) {
    DeviceRestore keep_device_;
    if (!idx || !queries || !ids || !out) return fail(ROAR_ERR_ARG, "bad arguments");
    if (!idx->base) return fail(ROAR_ERR_ARG, "distances before base vectors were loaded");
    if (nq == 0 || m == 0) return ROAR_OK;
    if (qstride_floats < idx->dim) return fail(ROAR_ERR_ARG, "query stride smaller than the dimension");
    for (uint64_t i = 0; i < nq * m; ++i)
        if (ids[i] >= idx->n) return fail(ROAR_ERR_ARG, "id %u outside the base set", ids[i]);
    std::lock_guard<std::mutex> guard(idx->mu);
    int e = use_device(idx);
    if (e) return e;
    float *q_dev = nullptr, *o_dev = nullptr;
    uint32_t *i_dev = nullptr;
    std::vector<float> buf;
    try {
        buf.assign((size_t)nq * idx->dim_pad, 0.0f);
    } catch (...) {
        return fail(ROAR_ERR_DEVICE, "out of host memory");
    }
    for (uint64_t i = 0; i < nq; ++i)
        memcpy(buf.data() + i * idx->dim_pad, queries + i * qstride_floats, idx->dim * sizeof(float));
    struct Free {  // every exit below releases the three device blocks
        float *&q, *&o;
        uint32_t *&i;
        ~Free() { (void)hipFree(q), (void)hipFree(o), (void)hipFree(i); }
    } free_all{q_dev, o_dev, i_dev};
    HIP_TRY(roar_dev_malloc((void **)&q_dev, buf.size() * sizeof(float)));
    HIP_TRY(roar_dev_malloc((void **)&i_dev, (size_t)nq * m * 4));
    HIP_TRY(roar_dev_malloc((void **)&o_dev, (size_t)nq * m * 4));
    HIP_TRY(hipMemcpy(q_dev, buf.data(), buf.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(i_dev, ids, (size_t)nq * m * 4, hipMemcpyHostToDevice));
    RoarDistParams p;
    memset(&p, 0, sizeof p);
    p.base = idx->base;
    p.ld = idx->ld;
    p.dim = idx->dim_pad;
    p.queries = q_dev;
    p.qstride = idx->dim_pad;
    p.nq = (uint32_t)nq;
    p.ids = i_dev;
    p.m = m;
    p.out = o_dev;
    p.off_ck = 0;
    p.off_cid = 8 * 64;
    p.off_q = p.off_cid + 4 * 64;
    p.lds_bytes = p.off_q + round_up(4 * idx->dim_pad, 16);
    int rc = roar_launch_distances(&p, idx->ip ? 1 : 0, nullptr);
    if (rc != 0) return fail(ROAR_ERR_DEVICE, "distance kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, o_dev, (size_t)nq * m * 4, hipMemcpyDeviceToHost));
    return ROAR_OK;
}

int roar_debug_screen_keeps(roar_index *idx, const float *queries, uint64_t nq, uint32_t qstride_floats, const uint32_t *ids,
                            const float *thresholds, uint32_t m, uint8_t *keeps) {
    DeviceRestore keep_device_;
    if (!idx || !queries || !ids || !thresholds || !keeps) return fail(ROAR_ERR_ARG, "bad arguments");
    if (!idx->base) return fail(ROAR_ERR_ARG, "screen check before base vectors were loaded");
    if (nq == 0 || m == 0) return ROAR_OK;
    if (qstride_floats < idx->dim) return fail(ROAR_ERR_ARG, "query stride smaller than the dimension");
    if (!(idx->dim_pad == 128 || idx->dim_pad == 512 || (idx->ip && idx->dim_pad == 200)))
        return fail(ROAR_ERR_ARG, "no screening kernel for rows of %u floats with this metric", idx->dim_pad);
    for (uint64_t i = 0; i < nq * m; ++i)
        if (ids[i] >= idx->n) return fail(ROAR_ERR_ARG, "id %u outside the base set", ids[i]);
    std::lock_guard<std::mutex> guard(idx->mu);
    int e = use_device(idx);
    if (e) return e;
    ensure_screen(idx);
    if (!idx->screen) return fail(ROAR_ERR_DEVICE, "the screening copy could not be made");
    float *q_dev = nullptr, *c_dev = nullptr;
    uint32_t *i_dev = nullptr;
    std::vector<float> buf;
    try {
        buf.assign((size_t)nq * idx->dim_pad, 0.0f);
    } catch (...) {
        return fail(ROAR_ERR_DEVICE, "out of host memory");
    }
    for (uint64_t i = 0; i < nq; ++i)
        memcpy(buf.data() + i * idx->dim_pad, queries + i * qstride_floats, idx->dim * sizeof(float));
    struct Free {  // every exit below releases the three device blocks
        float *&q, *&c;
        uint32_t *&i;
        ~Free() { (void)hipFree(q), (void)hipFree(c), (void)hipFree(i); }
    } free_all{q_dev, c_dev, i_dev};
    HIP_TRY(roar_dev_malloc((void **)&q_dev, buf.size() * sizeof(float)));
    HIP_TRY(roar_dev_malloc((void **)&i_dev, (size_t)nq * m * 4));
    HIP_TRY(roar_dev_malloc((void **)&c_dev, (size_t)nq * m * 4));
    HIP_TRY(hipMemcpy(q_dev, buf.data(), buf.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(i_dev, ids, (size_t)nq * m * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c_dev, thresholds, (size_t)nq * m * 4, hipMemcpyHostToDevice));
    int rc = idx->ip ? roar_launch_screen_check_1(idx->screen, idx->screen_mu, idx->dim_pad, q_dev, idx->dim_pad, nq, i_dev, c_dev, m, nullptr)
                     : roar_launch_screen_check_0(idx->screen, nullptr, idx->dim_pad, q_dev, idx->dim_pad, nq, i_dev, c_dev, m, nullptr);
    if (rc != 0) return fail(ROAR_ERR_DEVICE, "screen check launch failed: %s", hipGetErrorString((hipError_t)rc));
    HIP_TRY(hipDeviceSynchronize());
    std::vector<float> verdict;
    try {
        verdict.resize((size_t)nq * m);
    } catch (...) {
        return fail(ROAR_ERR_DEVICE, "out of host memory");
    }
    HIP_TRY(hipMemcpy(verdict.data(), c_dev, (size_t)nq * m * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < verdict.size(); ++i) keeps[i] = verdict[i] != 0.0f ? 1 : 0;
    return ROAR_OK;
}

namespace {
// The f32 form (roar_knn.hip): every score by v_mfma_f32_32x32x2_f32.
int exact_knn_f32(roar_index *idx, const float *queries_dev, uint64_t nq, uint32_t qstride_floats, uint32_t k, uint32_t *ids_dev, float *dists_dev,
                  void *hip_stream) {
    uint32_t cap = 2048;
    while (cap < 64u * k && cap < 8192) cap <<= 1;
    const uint64_t chunk = std::min<uint64_t>(nq, 65536);
    void *cand = nullptr, *cursor = nullptr, *tau = nullptr, *bnorm = nullptr, *ovf = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(cand); (void)hipFree(cursor); (void)hipFree(tau); (void)hipFree(bnorm); (void)hipFree(ovf);
    };
    {
        hipError_t he = roar_dev_malloc(&cursor, chunk * 4);
        if (he == hipSuccess) he = roar_dev_malloc(&tau, chunk * 4);
        if (he == hipSuccess) he = roar_dev_malloc(&ovf, 4);
        if (he == hipSuccess && !idx->ip) he = roar_dev_malloc(&bnorm, idx->n * 4);
        if (he != hipSuccess) {
            cleanup();
            return fail(ROAR_ERR_DEVICE, "out of device memory for the exact kNN workspaces: %s", hipGetErrorString(he));
        }
    }
    for (;;) {
        if (roar_dev_malloc(&cand, chunk * cap * 8) != hipSuccess) {
            cleanup();
            return fail(ROAR_ERR_DEVICE, "out of device memory for the kNN candidate lists");
        }
        bool retry = false;
        for (uint64_t q0 = 0; q0 < nq && !retry; q0 += chunk) {
            const uint32_t cnt = (uint32_t)std::min<uint64_t>(chunk, nq - q0);
            uint32_t overflowed = 0;
            int rc = roar_knn_run(queries_dev + q0 * qstride_floats, qstride_floats, cnt, idx->base, idx->ld, idx->n,
                                  idx->dim_pad, idx->ip ? 1 : 0, k, cap, cand, cursor, tau, bnorm, ovf, ids_dev + q0 * k,
                                  dists_dev ? dists_dev + q0 * k : nullptr, &overflowed, hip_stream);
            if (rc != 0) {
                cleanup();
                return fail(ROAR_ERR_DEVICE, "exact kNN kernels failed: %s", hipGetErrorString((hipError_t)rc));
            }
            if (overflowed) retry = true;
        }
        if (!retry) break;
        (void)hipFree(cand);
        cand = nullptr;
        if (cap >= 8192) {
            cleanup();
            return fail(ROAR_ERR_DEVICE, "exact kNN candidate lists overflowed at the largest capacity");
        }
        cap <<= 1;
    }
    cleanup();
    return ROAR_OK;
}

// The bf16-screened form (roar_knn_bf16.hip): scores by v_mfma_f32_32x32x16_bf16 with a rigorous bound, the rows that can be among the
// top k re-ranked on the f32 rows in the reference's summation order.  *done = false: not applicable here (rows too long, a norm that is
// not finite, no memory for the bf16 copy) -- the caller runs the f32 form.
int exact_knn_bf16(roar_index *idx, const float *queries_dev, uint64_t nq, uint32_t qstride_floats, uint32_t k, uint32_t *ids_dev, float *dists_dev,
                   void *hip_stream, bool *done) {
    *done = false;
    if (idx->dim_pad > 512) return ROAR_OK;
    hipStream_t st = (hipStream_t)hip_stream;
    const uint32_t KQ = roar_knn_bf16_row_quads(idx->dim_pad);
    const uint64_t chunk = std::min<uint64_t>(nq, 65536), chunk_pad = (chunk + 255) / 256 * 256;
    void *bb = nullptr, *bnorm = nullptr, *xmax = nullptr, *qb = nullptr, *cand = nullptr, *cursor = nullptr, *tau = nullptr, *slack = nullptr,
         *thr = nullptr, *ovf = nullptr, *rr = nullptr;
    auto cleanup = [&]() {
        for (void *p_ : {bb, bnorm, xmax, qb, cand, cursor, tau, slack, thr, ovf, rr}) (void)hipFree(p_);
    };
    hipError_t he = roar_dev_malloc(&bb, (size_t)idx->n * KQ * 16);
    if (he == hipSuccess) he = roar_dev_malloc(&bnorm, (size_t)idx->n * 4);
    if (he == hipSuccess) he = roar_dev_malloc(&xmax, 4);
    if (he == hipSuccess) he = roar_dev_malloc(&qb, (size_t)chunk_pad * KQ * 16);
    for (void **w : {&cursor, &tau, &slack, &thr})
        if (he == hipSuccess) he = roar_dev_malloc(w, chunk * 4);
    if (he == hipSuccess) he = roar_dev_malloc(&ovf, 4);
    if (he == hipSuccess) he = roar_dev_malloc(&rr, 4);
    if (he != hipSuccess) {  // (the f32 form needs far less memory: let it try)
        (void)hipGetLastError();
        cleanup();
        return ROAR_OK;
    }
    uint32_t xbits = 0;
    he = hipMemsetAsync(xmax, 0, 4, st);
    if (he == hipSuccess) he = hipMemsetAsync(rr, 0, 4, st);
    if (he == hipSuccess)
        he = (hipError_t)roar_knn_bf16_convert(idx->base, idx->ld, idx->dim_pad, idx->n, idx->n, bb, (float *)bnorm, (uint32_t *)xmax, st);
    if (he == hipSuccess) he = hipMemcpyAsync(&xbits, xmax, 4, hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    if (he != hipSuccess) {
        cleanup();
        return fail(ROAR_ERR_DEVICE, "exact kNN: bf16 copy of the base rows: %s", hipGetErrorString(he));
    }
    float xmax2;
    memcpy(&xmax2, &xbits, 4);
    if (!(xmax2 <= 1.0e36f)) {  // a row that is not finite (or whose norm overflows): no bound -- the f32 form scores everything
        cleanup();
        return ROAR_OK;
    }
    uint32_t cap = 2048;
    while (cap < 64u * k && cap < 8192) cap <<= 1;
    for (;;) {
        if (roar_dev_malloc(&cand, chunk * cap * 8) != hipSuccess) {
            cleanup();
            return fail(ROAR_ERR_DEVICE, "out of device memory for the kNN candidate lists");
        }
        bool retry = false;
        for (uint64_t q0 = 0; q0 < nq && !retry; q0 += chunk) {
            const uint32_t cnt = (uint32_t)std::min<uint64_t>(chunk, nq - q0);
            uint32_t overflowed = 0;
            int rc = roar_knn_run_bf16(queries_dev + q0 * qstride_floats, qstride_floats, cnt, idx->base, idx->ld, idx->n, idx->dim_pad,
                                       idx->ip ? 1 : 0, k, cap, bb, (const float *)bnorm, qb, cand, cursor, tau, slack, thr, ovf,
                                       (uint32_t *)rr, ids_dev + q0 * k, dists_dev ? dists_dev + q0 * k : nullptr, &overflowed, hip_stream);
            if (rc != 0) {
                cleanup();
                return fail(ROAR_ERR_DEVICE, "exact kNN (bf16 screen) kernels failed: %s", hipGetErrorString((hipError_t)rc));
            }
            if (overflowed) retry = true;
        }
        if (!retry) break;
        (void)hipFree(cand);
        cand = nullptr;
        if (cap >= 16384) {  // (the selection and the re-rank keep a list in LDS: 8 bytes per entry)
            cleanup();
            return ROAR_OK;  // the bound admits more than fits: the f32 form
        }
        cap <<= 1;
        (void)hipMemsetAsync(rr, 0, 4, st);
    }
    uint32_t rows = 0;
    (void)hipMemcpy(&rows, rr, 4, hipMemcpyDeviceToHost);
    idx->last_knn_rerank_rows = rows;
    cleanup();
    *done = true;
    return ROAR_OK;
}
}  // namespace

int roar_exact_knn_device(roar_index *idx, const float *queries_dev, uint64_t nq, uint32_t qstride_floats, uint32_t k,
                          uint32_t *ids_dev, float *dists_dev, void *hip_stream) {
    DeviceRestore keep_device_;
    if (!idx || !queries_dev || !ids_dev) return fail(ROAR_ERR_ARG, "bad kNN arguments");
    if (!idx->base || idx->n == 0) return fail(ROAR_ERR_ARG, "exact kNN before base vectors were loaded");
    if (k == 0 || k > 128 || k > idx->n) return fail(ROAR_ERR_ARG, "k must be in 1..min(128, n)");
    if (qstride_floats < idx->dim_pad || (qstride_floats & 3))
        return fail(ROAR_ERR_ARG, "device queries need a stride >= %u floats that is a multiple of 4", idx->dim_pad);
    if (nq == 0) return ROAR_OK;
    int e = use_device(idx);
    if (e) return e;
    idx->last_knn_bf16 = 0;
    idx->last_knn_rerank_rows = 0;
    if (idx->knn_mode != 1) {
        bool done = false;
        if ((e = exact_knn_bf16(idx, queries_dev, nq, qstride_floats, k, ids_dev, dists_dev, hip_stream, &done))) return e;
        if (done) {
            idx->last_knn_bf16 = 1;
            return ROAR_OK;
        }
    }
    return exact_knn_f32(idx, queries_dev, nq, qstride_floats, k, ids_dev, dists_dev, hip_stream);
}

int roar_exact_knn(roar_index *idx, const float *queries, uint64_t nq, uint32_t qstride_floats, uint32_t k, uint32_t *ids,
                   float *dists) {
    DeviceRestore keep_device_;
    if (!idx || !queries || !ids) return fail(ROAR_ERR_ARG, "bad kNN arguments");
    if (qstride_floats < idx->dim) return fail(ROAR_ERR_ARG, "query stride smaller than the dimension");
    if (nq == 0) return ROAR_OK;
    int e = use_device(idx);
    if (e) return e;
    const uint64_t chunk = std::min<uint64_t>(nq, 1u << 18);
    float *q_dev = nullptr, *d_dev = nullptr;
    uint32_t *i_dev = nullptr;
    std::vector<float> buf;
    try {
        buf.resize((size_t)chunk * idx->dim_pad);
    } catch (...) {
        return fail(ROAR_ERR_DEVICE, "out of host memory");
    }
    auto cleanup = [&]() { (void)hipFree(q_dev); (void)hipFree(i_dev); (void)hipFree(d_dev); };
    {
        hipError_t he = roar_dev_malloc((void **)&q_dev, buf.size() * sizeof(float));
        if (he == hipSuccess) he = roar_dev_malloc((void **)&i_dev, (size_t)chunk * k * 4);
        if (he == hipSuccess) he = roar_dev_malloc((void **)&d_dev, (size_t)chunk * k * 4);
        if (he != hipSuccess) {
            cleanup();
            return fail(ROAR_ERR_DEVICE, "out of device memory for the exact kNN staging blocks: %s", hipGetErrorString(he));
        }
    }
    for (uint64_t q0 = 0; q0 < nq; q0 += chunk) {
        const uint64_t cnt = std::min<uint64_t>(chunk, nq - q0);
        std::fill(buf.begin(), buf.begin() + cnt * idx->dim_pad, 0.0f);
        for (uint64_t i = 0; i < cnt; ++i)
            memcpy(buf.data() + i * idx->dim_pad, queries + (q0 + i) * qstride_floats, idx->dim * sizeof(float));
        hipError_t he = hipMemcpy(q_dev, buf.data(), (size_t)cnt * idx->dim_pad * sizeof(float), hipMemcpyHostToDevice);
        if (he == hipSuccess) {
            e = roar_exact_knn_device(idx, q_dev, cnt, idx->dim_pad, k, i_dev, d_dev, nullptr);
            if (e) { cleanup(); return e; }
            he = hipMemcpy(ids + q0 * k, i_dev, (size_t)cnt * k * 4, hipMemcpyDeviceToHost);
        }
        if (he == hipSuccess && dists) he = hipMemcpy(dists + q0 * k, d_dev, (size_t)cnt * k * 4, hipMemcpyDeviceToHost);
        if (he != hipSuccess) { cleanup(); return fail(ROAR_ERR_DEVICE, "exact kNN copy failed: %s", hipGetErrorString(he)); }
    }
    cleanup();
    return ROAR_OK;
}

int roar_merge_topk_device(int device, uint32_t nshards, uint64_t nq, uint32_t k, const uint32_t *ids_dev,
                           const float *dists_dev, uint32_t *out_ids_dev, float *out_dists_dev, void *hip_stream) {
    DeviceRestore keep_device_;
    if (nshards == 0 || nshards > 32 || k == 0) return fail(ROAR_ERR_ARG, "merge needs 1..32 shards and k > 0");
    if (!ids_dev || !dists_dev || !out_ids_dev || !out_dists_dev) return fail(ROAR_ERR_ARG, "bad merge arguments");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(ROAR_ERR_DEVICE, "no HIP device available; this library has no CPU path");
    HIP_TRY(hipSetDevice(device));
    int rc = roar_launch_merge_topk(nshards, nq, k, ids_dev, dists_dev, out_ids_dev, out_dists_dev, hip_stream);
    if (rc != 0) return fail(ROAR_ERR_DEVICE, "merge kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return ROAR_OK;
}

int roar_rebase_ids_device(int device, uint32_t *ids_dev, uint64_t count, uint32_t base, void *hip_stream) {
    DeviceRestore keep_device_;
    if (!ids_dev && count) return fail(ROAR_ERR_ARG, "bad rebase arguments");
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0)
        return fail(ROAR_ERR_DEVICE, "no HIP device available; this library has no CPU path");
    HIP_TRY(hipSetDevice(device));
    int rc = roar_launch_rebase_ids(ids_dev, count, base, hip_stream);
    if (rc != 0) return fail(ROAR_ERR_DEVICE, "rebase kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return ROAR_OK;
}

int roar_queue_trace(int device, uint32_t capacity, uint32_t n_ops, const uint8_t *ops, const uint32_t *ids, const float *dists,
                     int pool_in_registers, uint32_t *snap) {
    DeviceRestore keep_device_;
    if (!ops || !ids || !dists || !snap || capacity == 0 || capacity > 64)
        return fail(ROAR_ERR_ARG, "queue trace: capacity 1..64 and non-null arrays");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(ROAR_ERR_DEVICE, "no HIP device available; this library has no CPU path");
    HIP_TRY(hipSetDevice(device));
    if (n_ops == 0) return ROAR_OK;
    const size_t row = 2 + 3 * (size_t)capacity;
    uint8_t *d_ops = nullptr;
    uint32_t *d_ids = nullptr, *d_snap = nullptr;
    float *d_d = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_ops); (void)hipFree(d_ids); (void)hipFree(d_d); (void)hipFree(d_snap); };
    hipError_t e = roar_dev_malloc((void **)&d_ops, n_ops);
    if (e == hipSuccess) e = roar_dev_malloc((void **)&d_ids, (size_t)n_ops * 4);
    if (e == hipSuccess) e = roar_dev_malloc((void **)&d_d, (size_t)n_ops * 4);
    if (e == hipSuccess) e = roar_dev_malloc((void **)&d_snap, (size_t)n_ops * row * 4);
    if (e == hipSuccess) e = hipMemcpy(d_ops, ops, n_ops, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_ids, ids, (size_t)n_ops * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_d, dists, (size_t)n_ops * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = (hipError_t)roar_launch_queue_trace(capacity, n_ops, d_ops, d_ids, d_d, pool_in_registers ? 1 : 0, d_snap, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(snap, d_snap, (size_t)n_ops * row * 4, hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return fail(ROAR_ERR_DEVICE, "queue trace failed: %s", hipGetErrorString(e));
    return ROAR_OK;
}

int roar_debug_fail_device_alloc_at(int64_t n) {
    const int64_t left = g_fail_alloc_in.exchange(n > 0 ? n : 0);
    return (int)std::min<int64_t>(std::max<int64_t>(left, 0), 0x7FFFFFFF);  // > 0: it was armed and has not fired
}

int roar_last_profile(roar_index *idx, uint64_t *out, uint64_t nq) {
    DeviceRestore keep_device_;
    if (!idx || !out) return fail(ROAR_ERR_ARG, "bad arguments");
    if (!idx->prof_dev || idx->prof_nq == 0) return fail(ROAR_ERR_ARG, "no instrumented search has run (option profile_phases)");
    if (nq > idx->prof_nq) return fail(ROAR_ERR_ARG, "the last instrumented launch had %llu queries", (unsigned long long)idx->prof_nq);
    int e = use_device(idx);
    if (e) return e;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, idx->prof_dev, (size_t)nq * 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return ROAR_OK;
}

int roar_set_traffic_counters_device(roar_index *idx, uint32_t *counters_dev, uint64_t capacity_queries) {
    if (!idx) return fail(ROAR_ERR_ARG, "bad arguments");
    if (counters_dev && ((uintptr_t)counters_dev & 15u)) return fail(ROAR_ERR_ARG, "the counter block must be 16-byte aligned");
    std::lock_guard<std::mutex> guard(idx->mu);
    idx->traffic_dev = counters_dev;
    idx->traffic_cap = counters_dev ? capacity_queries : 0;
    return ROAR_OK;
}

int roar_set_option(roar_index *idx, const char *key, int64_t value) {
    if (!idx || !key) return fail(ROAR_ERR_ARG, "bad arguments");
    std::string k(key);
    if (k == "visited_mode") {
        if (value != ROAR_VISITED_EXACT && value != ROAR_VISITED_LDS && value != ROAR_VISITED_AUTO && value != ROAR_VISITED_LDS_FORGETFUL)
            return fail(ROAR_ERR_ARG, "bad visited_mode");
        idx->visited_mode = (int)value;
    } else if (k == "waves_per_cu") {
        if (value < 0 || value > 32) return fail(ROAR_ERR_ARG, "waves_per_cu must be 0..32");
        idx->waves_per_cu = (int)value;
    } else if (k == "vis_entries") {
        {
            int64_t v = value;
            while (v > 3 && !(v & 1)) v >>= 1;
            if (value < 0 || (value > 0 && v != 1 && v != 2 && v != 3)) return fail(ROAR_ERR_ARG, "vis_entries must be 0, 2^k or 3 * 2^k");
        }
        idx->vis_entries = (int)value;
    } else if (k == "tag_protect") {
        if (value < -1 || value > 1) return fail(ROAR_ERR_ARG, "tag_protect must be -1 (measured), 0 or 1");
        if (idx->tag_protect != (int)value) idx->plan_cache.clear();  // (table sizes were measured with the other setting)
        idx->tag_protect = (int)value;
    } else if (k == "search_kernel") {
        if (value < 0 || value > 3) return fail(ROAR_ERR_ARG, "search_kernel must be 0 (auto), 1, 2 or 3");
        if (idx->search_kernel != (int)value) idx->plan_cache.clear();  // (another kernel form: measured choices no longer apply)
        idx->search_kernel = (int)value;
    } else if (k == "hub_mode") {
        if (value < 0 || value > 2) return fail(ROAR_ERR_ARG, "hub_mode must be 0 (auto), 1 (off) or 2 (on)");
        if (idx->hub_mode != (int)value) idx->plan_cache.clear();  // (another kernel form: measured choices no longer apply)
        idx->hub_mode = (int)value;
    } else if (k == "launch_events") {
        idx->launch_events = value ? 1 : 0;
    } else if (k == "knn_mode") {
        if (value != 0 && value != 1) return fail(ROAR_ERR_ARG, "knn_mode must be 0 (auto: bf16 screen + exact re-rank) or 1 (f32 MFMA only)");
        idx->knn_mode = (int)value;
    } else if (k == "calibrate") {
        if (value < 0 || value > 2) return fail(ROAR_ERR_ARG, "calibrate must be 0 (defaults), 1 (measure, default) or 2 (forget what was measured)");
        if (value == 2) idx->plan_cache.clear();
        else idx->calibrate = (int)value;
    } else if (k == "screen") {
        if (value < 0 || value > 2) return fail(ROAR_ERR_ARG, "screen must be 0 (auto), 1 (off) or 2 (on)");
        idx->screen_mode = (int)value;
    } else if (k == "pool_form") {
        if (value != 0 && value != 1) return fail(ROAR_ERR_ARG, "pool_form must be 0 (auto) or 1 (LDS pool)");
        if (idx->pool_form != (int)value) idx->plan_cache.clear();  // (another kernel form: measured choices no longer apply)
        idx->pool_form = (int)value;
    } else if (k == "profile_phases") {
        idx->profile = value ? 1 : 0;
    } else if (k == "combine_single_queries") {
        idx->combine = value ? 1 : 0;
    } else if (k == "bipartite_seed") {
        idx->bp_seed = (uint64_t)value;
    } else if (k == "max_grid") {
        if (value < 0) return fail(ROAR_ERR_ARG, "max_grid must be >= 0");
        idx->max_grid = (int)value;
    } else {
        return fail(ROAR_ERR_ARG, "Invalid parameter name.");  // parameters.h:28
    }
    return ROAR_OK;
}

int roar_get_info(roar_index *idx, const char *key, int64_t *value) {
    DeviceRestore keep_device_;
    if (!idx || !key || !value) return fail(ROAR_ERR_ARG, "bad arguments");
    std::string k(key);
    if (k == "n") *value = (int64_t)idx->n;
    else if (k == "dim") *value = idx->dim;
    else if (k == "dim_padded") *value = idx->dim_pad;
    else if (k == "ep") *value = idx->ep;
    else if (k == "max_degree") *value = idx->max_deg;
    else if (k == "nedges") *value = (int64_t)idx->nedges;
    else if (k == "adj_stride") *value = idx->adj_stride;
    else if (k == "row_stride") *value = idx->ld;
    else if (k == "n_cus") *value = idx->n_cus;
    else if (k == "need_normalize") *value = idx->need_normalize ? 1 : 0;
    else if (k == "visited_mode") *value = idx->visited_mode;
    else if (k == "last_visited_mode") *value = idx->last_mode;
    else if (k == "last_visited_forgetful") *value = idx->last_forgetful;
    else if (k == "last_waves_per_query")
        *value = idx->last_sel.form == ROAR_FORM_BIPARTITE_WG ? 8
                 : idx->last_sel.form == ROAR_FORM_SEARCH ? idx->last_sel.waves : 1;
    else if (k == "last_register_pool") *value = idx->last_sel.regpool;
    else if (k == "last_hub_bitmap") *value = idx->last_sel.hub;
    else if (k == "last_screen") *value = idx->last_screen;
    else if (k == "last_tag_protect") *value = idx->last_tag_protect;
    else if (k == "last_plan_calibrated") *value = idx->last_calibrated;
    else if (k == "last_knn_bf16") *value = idx->last_knn_bf16;
    else if (k == "last_knn_rerank_rows") *value = idx->last_knn_rerank_rows;
    else if (k == "calibrations") *value = (int64_t)idx->plan_cache.size();
    else if (k == "screen_row_bytes") *value = roar_screen8_stride(idx->dim_pad);
    else if (k == "launch_events_on") *value = idx->launch_events;
    else if (k == "last_combined") *value = idx->last_combined;
    else if (k == "bipartite_nodes") *value = (int64_t)idx->bp_n;
    else if (k == "bipartite_edges") *value = (int64_t)idx->bp_edges;
    else if (k == "graph_hash") {
        if (!idx->adj || idx->gn == 0) return fail(ROAR_ERR_ARG, "no graph loaded");
        if (!idx->graph_hash_valid) {
            HIP_TRY(hipSetDevice(idx->device));
            unsigned long long *h_dev = nullptr, h = 0;
            HIP_TRY(roar_dev_malloc((void **)&h_dev, 8));
            int rc = roar_bk_graph_hash(idx->adj, idx->adj_stride, idx->gn, idx->ep, h_dev, nullptr);
            hipError_t he = rc == 0 ? hipMemcpy(&h, h_dev, 8, hipMemcpyDeviceToHost) : (hipError_t)rc;
            (void)hipFree(h_dev);
            if (he != hipSuccess) return fail(ROAR_ERR_DEVICE, "graph hash failed: %s", hipGetErrorString(he));
            idx->graph_hash = h;
            idx->graph_hash_valid = true;
        }
        *value = (int64_t)idx->graph_hash;
    }
    else if (k == "last_grid") *value = idx->last_grid;
    else if (k == "last_lds_bytes") *value = idx->last_lds;
    else if (k == "last_vis_entries") *value = idx->last_vis_entries;
    else if (k == "last_kernel_ns") {
        if (!idx->ev_valid) return fail(ROAR_ERR_ARG, "no search has been launched yet");
        HIP_TRY(hipSetDevice(idx->device));
        HIP_TRY(hipEventSynchronize(idx->ev1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, idx->ev0, idx->ev1));
        *value = (int64_t)((double)ms * 1e6);
    } else {
        return fail(ROAR_ERR_ARG, "Invalid parameter name.");
    }
    return ROAR_OK;
}

}  // extern "C"
