// roar_internal.h -- what the host-side translation units of libroargraph_amd.so share: the index object, launch plans, error and
// allocation helpers, and the declarations of the kernel launchers the .hip files export.  Internal; the public boundary is
// include/roargraph_amd.h.  (Round 5: these lived at the top of a 3 203-line roar_capi.cpp.)
//
//   roar_capi.cpp      the index object: errors, device buffers, base rows, graph files, visited pool
//   roar_plan.cpp      launch plans (plan_search) and their calibration on the loaded index (resolve_plan)
//   roar_launch.cpp    roar_search_batch_device: launch cell, kernel launch, ordering of launches that share HBM lists
//   roar_stream.cpp    roar_search_stream_*: a stream of batches served by persistent workgroups (round 5)
//   roar_host.cpp      host-buffer calls: staging, packing behind a running launch, call combining, replicas
//   roar_buildapi.cpp  graph builders          roar_bipartite.cpp  the two-hop variant
//   roar_knnapi.cpp    exact kNN generator     roar_misc.cpp       distances, debug surfaces, options, info keys
#pragma once
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
                          int aug, void *stream);
int roar_knn_run_bf16(const float *q, uint64_t ldq, uint32_t nq, const float *b, uint64_t ldb, uint64_t n, uint32_t dim, int ip, uint32_t k,
                      uint32_t cap, const void *bb, const float *bnorm, void *qb, void *cand, void *cursor, void *tau, void *ea, void *kq,
                      void *overflow_dev, uint32_t *rerank_rows_dev, uint32_t *ids, float *dists, uint32_t *overflowed, void *stream);
}

struct roar_index;

namespace roar_impl {

extern thread_local std::string g_err;  // the message behind roar_last_error()
int fail(int code, const char *fmt, ...);

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(ROAR_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

inline uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

// Every device allocation goes through here so that error paths can be exercised (roar_debug_fail_device_alloc_at); unarmed it is hipMalloc.
extern std::atomic<int64_t> g_fail_alloc_in;
hipError_t roar_dev_malloc(void **p, size_t bytes);
template <typename T>
hipError_t roar_dev_malloc(T **p, size_t bytes) {
    return roar_dev_malloc(reinterpret_cast<void **>(p), bytes);
}

}  // namespace roar_impl

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
    // "edge floats": the neighbours' screening floats beside the adjacency (RoarSearchParams::adj_meta), made on first use from the graph
    // and the screening copy, dropped when either changes
    unsigned long long *adj_meta = nullptr;
    int edge_floats_mode = 0;  // option "edge_floats": 0 auto (on where the kernels have the form), 1 off
    int last_ann = 0;
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
        uint32_t wpc = 0;  // workgroups per CU the winning plan kept resident when it was measured (0: not recorded)
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
    int last_wpc = 0;  // workgroups per CU of the last plan (before the grid is cut to the batch or a cap)
    std::vector<std::pair<uint64_t, int>> occ_cache;  // (kernel form, LDS bytes) -> resident workgroups per CU
    std::atomic<int> streams_open{0};  // roar_search_stream_open .. _close: base rows and graph must stay as they are meanwhile
};

namespace roar_impl {

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

constexpr uint32_t LAUNCH_RING = 256;  // launch cells: more launches than this are never in flight on one index
constexpr size_t ADJ_SLACK_WORDS = 256;  // a row read is up to 4 x 64 words from the row's first word

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
    int stream = 0;        // 1: plan the batch-stream form of the kernel (roar_stream.cpp): one wave per query, forgetful on-chip table
};

// does this index have a screening kernel for its rows (roar_search_v2.h: 128-d and 512-d rows, 200-d inner product)
inline bool has_screen_kernel(const roar_index *idx) {
    return idx->dim_pad == 128 || idx->dim_pad == 512 || (idx->ip && idx->dim_pad == 200);
}

// roar_capi.cpp
int use_device(roar_index *idx);
int refuse_while_streaming(const roar_index *idx);  // base rows / graph are about to change: not while a search stream is open
void free_graph(roar_index *idx);
void free_bipartite(roar_index *idx);
void free_base(roar_index *idx);
void free_tags(roar_index *idx);
void normalize_row(float *arr, size_t dim);
int alloc_base(roar_index *idx, uint64_t n);
int upload_rows(roar_index *idx, const float *padded, uint64_t r0, uint64_t cnt);
uint32_t affordable_slots(uint64_t bytes_per_slot, uint64_t already_held, uint32_t want);
int ensure_tags(roar_index *idx, uint32_t *slots_io);
int ensure_spill(roar_index *idx, uint32_t slots, bool *granted);
int ensure_tags32(roar_index *idx, uint32_t *slots_io);
int ensure_host_csr(roar_index *idx);
int ensure_hub_rows(roar_index *idx);
void ensure_screen(roar_index *idx);
void ensure_edge_floats(roar_index *idx);  // (failure to make them is not an error: adj_meta stays null and the launch reads the floats from the copy)
void free_edge_floats(roar_index *idx);
void make_screen_with_base(roar_index *idx);

// roar_plan.cpp
// LDS carve-up and grid for one search launch.  `pick`: measured choices for what no option fixes (null: defaults).
int plan_search(roar_index *idx, uint64_t nq, uint32_t k, uint32_t L, LaunchPlan *plan, bool bipartite = false, const PlanPick *pick = nullptr);
// The plan of a launch: calibrated choices where there are any (measuring them now if `qdev` is given and this kind of launch
// has not been seen yet), defaults otherwise.  *need_queries (may be null): set when a calibration is due but qdev is null.
int resolve_plan(roar_index *idx, const float *qdev, uint64_t nq, uint32_t qstride, uint32_t k, uint32_t L, hipStream_t st, LaunchPlan *plan,
                 bool *need_queries);

// The plan of a stream's workgroups (roar_stream.cpp): the batch-stream form of the kernel, with what the calibration measured for
// the largest batches of this L_pq (measuring it now on `qdev` if need be); plan->grid = the workgroups the chip holds at once.
int resolve_stream_plan(roar_index *idx, const float *qdev, uint64_t nq, uint32_t qstride, uint32_t k, uint32_t L, hipStream_t st, LaunchPlan *plan,
                        bool *provisional);
uint64_t plan_key(const roar_index *idx, uint64_t nq, uint32_t L);  // which cached calibration serves a launch like this one
const roar_index::Calibration *find_calibration(const roar_index *idx, uint64_t key);
uint64_t stream_calibration_min_queries(const roar_index *idx);  // a smaller sample cannot make the calibration of the largest batches

// roar_launch.cpp
int launch_planned(roar_index *idx, LaunchPlan &plan, hipStream_t st);
int search_device(roar_index *idx, const float *queries_dev, uint64_t nq, uint32_t qstride_floats, uint32_t k, uint32_t L_pq,
                  uint32_t *ids_dev, float *dists_dev, uint32_t *cmps_dev, uint32_t *hops_dev, uint32_t *scanned_dev,
                  uint32_t *short_count_dev, void *hip_stream, const uint32_t *ready_dev, const LaunchPlan *planned = nullptr);

// roar_host.cpp
int stage_queries(roar_index *idx, const float *queries, uint64_t nq, uint32_t qstride_floats, uint32_t k, const float **qdev,
                  uint32_t *qstride, bool *deferred);
void pack_queries(roar_index *idx, const float *queries, uint64_t nq, uint32_t qstride_floats);
int fetch_results(roar_index *idx, uint64_t nq, uint32_t k, uint32_t *ids, float *dists, uint32_t *cmps, uint32_t *hops,
                  uint32_t *scanned);
int search_host(roar_index *idx, const float *queries, uint64_t nq, uint32_t qstride_floats, uint32_t k, uint32_t L_pq,
                uint32_t *ids, float *dists, uint32_t *cmps, uint32_t *hops, uint32_t *scanned);

}  // namespace roar_impl
