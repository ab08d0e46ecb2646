// roar_capi.cpp -- C ABI (include/roargraph_amd.h): the index object -- errors, device buffers, base rows, graph files, visited pool.  Host-only code built with hipcc.  There is deliberately no CPU compute path in this library.
// (round 5: roar_capi.cpp was one 3 203-line translation unit; it is now cut by concern.  Shared declarations: roar_internal.h.)
#include "roar_internal.h"

using namespace roar_impl;

// Launches of one index that overlap on several HIP streams need as many hardware queues as streams: a launch starts only when the
// launch before it in the same hardware queue has finished, and the runtime maps streams onto four queues unless GPU_MAX_HW_QUEUES
// says otherwise WHEN IT STARTS.  Set (never overwritten) when this library is loaded -- before the HIP runtime's first call in a
// process that links it ahead of any HIP use; a process whose runtime is already up is unaffected (info key "hw_queues_env").
namespace {
struct HwQueuesEnv {
    HwQueuesEnv() { (void)setenv("GPU_MAX_HW_QUEUES", "8", 0); }
} g_hw_queues_env;
}  // namespace

namespace roar_impl {

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
}  // namespace roar_impl

namespace roar_impl {

int use_device(roar_index *idx) {
    HIP_TRY(hipSetDevice(idx->device));
    return ROAR_OK;
}

int refuse_while_streaming(const roar_index *idx) {
    if (idx && idx->streams_open.load() > 0)
        return fail(ROAR_ERR_ARG, "a search stream is open on this index (roar_search_stream_open): close it before replacing base rows or graph");
    return ROAR_OK;
}

void free_edge_floats(roar_index *idx) {
    if (idx->adj_meta) (void)hipFree(idx->adj_meta);
    idx->adj_meta = nullptr;
}

void free_graph(roar_index *idx) {
    idx->plan_cache.clear();  // (measured on the graph that goes away)
    free_edge_floats(idx);
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
    free_edge_floats(idx);
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

// The screening copy (roar_search_v2.h, screen8_pass): + roar_screen8_bytes(dim, n) of HBM (256 per row at 200-d, 512 + 16 at 512-d), one
// conversion pass.  Failure to make it is not an
// error: the launch then scores every neighbour on the f32 rows, as without it.
// RoarSearchParams::adj_meta: n x adjacency stride x 8 bytes (WebVid-2.5M shape: 2.6 GB), one pass over the adjacency.
void ensure_edge_floats(roar_index *idx) {
    if (idx->adj_meta || !idx->screen || !idx->adj || idx->gn != idx->n || !idx->ip || !roar_screen8_split(idx->dim_pad)) return;
    const size_t words = (size_t)idx->gn * idx->adj_stride + ADJ_SLACK_WORDS;
    if (roar_dev_malloc((void **)&idx->adj_meta, words * 8) != hipSuccess) {
        (void)hipGetLastError();
        idx->adj_meta = nullptr;
        return;
    }
    hipError_t e = hipMemset(idx->adj_meta, 0, words * 8);
    const void *floats = idx->screen + (size_t)idx->n * roar_screen8_stride(idx->dim_pad);
    if (e == hipSuccess) e = (hipError_t)roar_launch_edge_floats(idx->adj, idx->adj_stride, idx->gn, floats, idx->adj_meta, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipGetLastError();
        free_edge_floats(idx);
    }
}

void ensure_screen(roar_index *idx) {
    if (idx->screen && idx->screen_n == idx->n) return;
    free_edge_floats(idx);  // (made from the copy that is being replaced)
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
    if (roar_dev_malloc((void **)&idx->screen, (size_t)roar_screen8_bytes(idx->dim_pad, idx->n) + 256) != hipSuccess) return drop();
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
}  // namespace roar_impl

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
        HIP_TRY(roar_dev_malloc((void **)&idx->ring, LAUNCH_RING * 5 * sizeof(uint32_t)));
        HIP_TRY(hipMemset(idx->ring, 0, LAUNCH_RING * 5 * sizeof(uint32_t)));
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
    if (idx && idx->streams_open.load() > 0) {  // (workgroups may still be reading the index: leak it rather than free it under them)
        (void)fail(ROAR_ERR_ARG, "roar_destroy: a search stream is still open on this index; the index was NOT freed");
        return;
    }
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
    if (int busy_ = refuse_while_streaming(idx)) return busy_;
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
    if (int busy_ = refuse_while_streaming(idx)) return busy_;
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
    if (int busy_ = refuse_while_streaming(idx)) return busy_;
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
    if (int busy_ = refuse_while_streaming(idx)) return busy_;
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
    if (int busy_ = refuse_while_streaming(idx)) return busy_;
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

int roar_init_visited_pool(roar_index *idx, uint32_t slots) {
    DeviceRestore keep_device_;
    if (!idx || idx->n == 0) return fail(ROAR_ERR_ARG, "InitVisitedListPool before the base vectors were loaded");
    int e = use_device(idx);
    if (e) return e;
    slots = std::max<uint32_t>(slots, 1);
    return ensure_tags(idx, &slots);
}
}  // extern "C"
