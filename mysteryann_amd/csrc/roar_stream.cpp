// roar_stream.cpp -- roar_search_stream_{open,submit,wait,close}: a STREAM of query batches served by workgroups that outlive a batch.
//
// What it replaces: the harness loop tests/test_search_roargraph.cpp:202-212 (`omp parallel for schedule(dynamic, 1)` over the queries
// of one batch) when batches arrive one after another -- one GPU's share of a batch at N = 8 is 1 250 queries, and a launch per
// batch ends with its slowest query (75-89 hops against 56 on average) while most of the chip idles.  Here a batch is not a launch:
//
//   * the host PUBLISHES a batch: a 64-byte slot {query block, result arrays, nq} in a ring in pinned host memory, then `tail`;
//   * workgroups of the batch-stream form of the search kernel (roar_search_v2.h, SRC == 2) take queries from the published batches in
//     batch order -- one atomicAdd on the batch's counter cell per query, as the per-launch form does on its launch cell -- cross
//     batch boundaries, and leave only when nothing published is left;
//   * the workgroup that finishes a batch's last query (a second counter in the cell) writes the batch's completion record to pinned
//     host memory; roar_search_stream_wait polls it.  Results are bit-identical to separate launches (the same search code).
//
// Nothing spins waiting for the host: a workgroup that finds nothing published counts itself out and exits, so a stream that is open
// but idle holds no part of the chip, and no kernel can outlive the work it was given.  submit() therefore tops the workgroups up: it
// knows how many it has launched, reads how many have left (every leaving workgroup posts the running count to pinned memory), and
// launches min(room on the chip, queries nobody has taken yet) more on one of its own HIP streams.  The race "the last workgroup
// leaves while a batch is being published" is closed on both sides (publish, THEN read `exited`; post the count, THEN look at the
// host's `tail` once more) and, belt and braces, wait() starts workgroups itself when it finds a batch pending and nobody -- or too few
// -- alive.
//
// PCIe is kept out of the workgroups' way (roar_device.h): a batch's slot is fetched from pinned memory by ONE workgroup and mirrored in
// device memory for the others, the host's `tail` is read by one workgroup at a time and mirrored, results and counters live in device
// memory; per batch the bus carries two reads and four small posted writes.
//
// Counter cells are indexed by batch NUMBER and never reused while a workgroup may be alive (a late workgroup may touch the cell of a
// long finished batch): after `cap` batches submit() drains the stream -- every batch complete, every launch finished -- and zeroes
// them.  Slots are reused every `ring` batches, only after the batch that held them is complete; a workgroup checks the slot's batch
// number (seqlock) and treats a mismatch as "that batch was complete long ago".
#include <chrono>
#include <thread>
#include <unordered_map>

#include "roar_internal.h"

using namespace roar_impl;

struct roar_stream {
    roar_index *idx = nullptr;
    uint32_t k = 0, L = 0;
    uint32_t ring = 256;   // slots (power of two): batches that may be incomplete at once
    uint32_t cap = 65536;  // batch numbers per epoch (counter cells)
    bool planned = false;
    bool provisional = false;  // planned on defaults: no batch so far was large enough to calibrate on (submit() plans again when one is)
    LaunchPlan plan;
    uint32_t R = 0;  // workgroups the chip holds at once (plan.grid)
    // pinned host memory (one block) and its device address
    unsigned char *pin = nullptr, *pin_dev = nullptr;
    RoarStreamCtl *ctl = nullptr;
    RoarStreamSlot *slots = nullptr;
    RoarStreamDone *done = nullptr;
    RoarStreamCell *cells = nullptr;  // device memory
    RoarStreamDev *dev = nullptr;     // device memory: what the workgroups share (mirrors of the host's tail and slots)
    uint32_t max_exited = 0;          // largest RoarStreamCtl::exited seen (posted writes of several workgroups may land out of order)
    uint32_t submitted = 0;   // batch numbers handed out in this epoch
    uint32_t completed = 0;   // every batch < completed is known complete
    uint64_t pending_q = 0;   // queries of the batches in [completed, submitted)
    uint32_t claimed = 0;     // every batch < claimed has handed out all its queries (they are finished or held by a live workgroup)
    uint64_t unclaimed_q = 0; // queries of the batches in [claimed, submitted): an upper bound of what no workgroup has taken yet
    uint64_t epoch_base = 0;  // ticket = epoch_base + batch number + 1
    uint64_t launched = 0;    // workgroups launched in this epoch
    std::vector<uint32_t> nq_of;      // per batch number of the epoch
    // per-ticket records (short count, device wall clock of the first query taken / the batch complete), index ticket - 1 - rec_base: those
    // of the running epoch and of the one before it.  When an epoch ends the older ones are dropped (a stream serves for days): of a
    // dropped ticket only a NON-ZERO short count is kept (old_short: rare), so wait() answers for every ticket ever handed out.
    std::vector<uint32_t> short_of;
    std::vector<uint64_t> first_of, done_of;
    uint64_t rec_base = 0;
    std::unordered_map<uint64_t, uint32_t> old_short;
    struct Lane {
        hipStream_t st = nullptr;
        hipEvent_t ev = nullptr;
        bool used = false;
    };
    std::vector<Lane> lanes;
    std::mutex mu;
    int64_t n_launches = 0, n_heals = 0, n_resets = 0, n_replans = 0;
    int wall_khz = 100000;
    bool broken = false;  // a launch failed or a wait timed out: every later call fails
};

namespace {

constexpr double WAIT_TIMEOUT_S = 60.0;
constexpr double CLOSE_GRACE_S = 10.0;  // close() of a failed stream: how long its launches are given to end (device side: 2 s)

size_t pin_bytes(const roar_stream *s) { return sizeof(RoarStreamCtl) + (size_t)s->ring * (sizeof(RoarStreamSlot) + sizeof(RoarStreamDone)); }

size_t dev_bytes(const roar_stream *s) { return sizeof(RoarStreamDev) + (size_t)s->cap * sizeof(RoarStreamSlot); }

uint32_t alive(roar_stream *s) {
    const uint32_t ex = *reinterpret_cast<volatile uint32_t *>(&s->ctl->exited);
    if (ex > s->max_exited && (uint64_t)ex <= s->launched) s->max_exited = ex;
    return (uint32_t)(s->launched - (uint64_t)s->max_exited);
}

// g more workgroups, on a HIP stream of the stream's own whose last launch has finished (launches on one HIP stream run one after the
// other, and a launch lives as long as there is work)
int launch_wgs(roar_stream *s, uint32_t g) {
    if (g == 0) return ROAR_OK;
    roar_stream::Lane *ln = nullptr;
    for (auto &c : s->lanes) {
        if (!c.used) {
            ln = &c;
            break;
        }
        const hipError_t q = hipEventQuery(c.ev);
        if (q == hipSuccess) {
            ln = &c;
            break;
        }
        if (q != hipErrorNotReady) {
            s->broken = true;
            return fail(ROAR_ERR_DEVICE, "search stream: a launch failed: %s", hipGetErrorString(q));
        }
        (void)hipGetLastError();
    }
    if (!ln) {
        if (s->lanes.size() >= 64) ln = &s->lanes[s->n_launches % s->lanes.size()];  // (all busy: queue behind one of them)
        else {
            roar_stream::Lane c;
            HIP_TRY(hipStreamCreateWithFlags(&c.st, hipStreamNonBlocking));
            if (hipEventCreateWithFlags(&c.ev, hipEventDisableTiming) != hipSuccess) {
                (void)hipStreamDestroy(c.st);
                return fail(ROAR_ERR_DEVICE, "search stream: cannot create an event");
            }
            try {
                s->lanes.push_back(c);
            } catch (...) {
                (void)hipEventDestroy(c.ev);
                (void)hipStreamDestroy(c.st);
                return fail(ROAR_ERR_DEVICE, "out of host memory");
            }
            ln = &s->lanes.back();
        }
    }
    s->plan.p.s_tail = s->submitted;
    s->plan.p.s_start = s->claimed;  // (every query of the batches before it has been handed out: nothing there for a new workgroup)
    const int rc = roar_launch_search(&s->plan.p, &s->plan.sel, g, ln->st);
    if (rc != 0) {
        s->broken = true;
        return fail(ROAR_ERR_DEVICE, "search stream: kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    }
    HIP_TRY(hipEventRecord(ln->ev, ln->st));
    ln->used = true;
    s->launched += g;
    s->n_launches++;
    return ROAR_OK;
}

// completion records -> host bookkeeping (in batch order: `completed` is a prefix)
void poll(roar_stream *s) {
    while (s->completed < s->submitted) {
        const uint32_t b = s->completed;
        volatile RoarStreamDone *d = &s->done[b & (s->ring - 1)];
        if (d->seq1_done != b + 1u) break;
        std::atomic_thread_fence(std::memory_order_acquire);
        const uint64_t t = s->epoch_base + b - s->rec_base;
        s->short_of[t] = d->nshort;
        s->first_of[t] = d->t_first;
        s->done_of[t] = d->t_done;
        s->pending_q -= s->nq_of[b];
        s->completed = b + 1;
    }
    while (s->claimed < s->submitted) {
        const uint32_t b = s->claimed;
        volatile RoarStreamDone *d = &s->done[b & (s->ring - 1)];
        if (b >= s->completed && d->seq1_claimed != b + 1u) break;
        s->unclaimed_q -= s->nq_of[b];
        s->claimed = b + 1;
    }
}

bool batch_done(roar_stream *s, uint32_t b) {  // (a batch may finish before an earlier one)
    if (b < s->completed) return true;
    volatile RoarStreamDone *d = &s->done[b & (s->ring - 1)];
    return d->seq1_done == b + 1u;
}

// workgroups for what nobody has taken yet: up to the chip's capacity, never more than there are such queries (known per batch: the
// workgroup that takes a batch's last query says so in the batch's record)
// (from wait(): only when nobody is alive, or when a sixteenth of the chip could be put to work -- the host's view of who is alive lags)
int top_up(roar_stream *s, bool from_wait) {
    const uint32_t live = alive(s);
    const uint64_t room = s->R > live ? s->R - live : 0;
    uint64_t g = std::min<uint64_t>(room, s->unclaimed_q);
    if (from_wait) {
        if (live == 0 && s->pending_q) g = std::max<uint64_t>(g, 1);
        else if (g < std::max<uint32_t>(s->R / 16, 1)) g = 0;
    }
    if (g == 0) return ROAR_OK;
    if (from_wait) s->n_heals++;
    return launch_wgs(s, (uint32_t)g);
}

// block until `pred` holds; workgroups are started if a batch is pending and nobody is alive
// (the stream's mutex is held on entry and on return, and released between looks: another thread may submit meanwhile)
template <typename Pred>
int wait_for(roar_stream *s, std::unique_lock<std::mutex> &lk, Pred pred, const char *what) {
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t spins = 0;
    for (;;) {
        poll(s);
        if (pred()) return ROAR_OK;
        if (s->broken) return fail(ROAR_ERR_DEVICE, "search stream: an earlier call failed; close the stream");
        if (const uint32_t code = *reinterpret_cast<volatile uint32_t *>(&s->ctl->error)) {
            s->broken = true;
            return fail(ROAR_ERR_DEVICE, "search stream: a workgroup gave up waiting for another one (code %u: 1 slot fetch, 2 batch count, 3 tail look)", code);
        }
        if ((++spins & 0x3FFu) == 0) {
            if (s->pending_q) {
                int e = top_up(s, true);
                if (e) return e;
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > WAIT_TIMEOUT_S) {
                s->broken = true;
                return fail(ROAR_ERR_DEVICE, "search stream: %s did not happen within %.0f s (%u of %u batches complete, %u workgroups alive)", what,
                            WAIT_TIMEOUT_S, s->completed, s->submitted, alive(s));
            }
        }
        lk.unlock();
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
        lk.lock();
    }
}

// the plan of the stream's workgroups, on this batch's own queries (a calibration, if one is due, runs ordinary launches here)
int plan_stream(roar_stream *s, const float *queries_dev, uint64_t nq, uint32_t qstride_floats) {
    roar_index *idx = s->idx;
    hipStream_t st = idx->stream;
    {
        std::lock_guard<std::mutex> g2(idx->mu);
        int e = resolve_stream_plan(idx, queries_dev, nq, qstride_floats, s->k, s->L, st, &s->plan, &s->provisional);
        if (e) return e;
        HIP_TRY(hipStreamSynchronize(st));
    }
    RoarSearchParams &p = s->plan.p;
    p.s_ctl = reinterpret_cast<const RoarStreamCtl *>(s->pin_dev);
    p.s_slots = reinterpret_cast<const RoarStreamSlot *>(s->pin_dev + sizeof(RoarStreamCtl));
    p.s_done = reinterpret_cast<RoarStreamDone *>(s->pin_dev + sizeof(RoarStreamCtl) + (size_t)s->ring * sizeof(RoarStreamSlot));
    p.s_cells = s->cells;
    p.s_dev = s->dev;
    p.s_ring = s->ring;
    p.nq = 0;
    s->R = std::max<uint32_t>(s->plan.grid, 1);
    s->planned = true;
    return ROAR_OK;
}

// every batch complete, every launch finished, counters zero: a new epoch of batch numbers
int reset_epoch(roar_stream *s, std::unique_lock<std::mutex> &lk) {
    int e = wait_for(s, lk, [&] { return s->completed == s->submitted; }, "the drain of the stream");
    if (e) return e;
    for (auto &c : s->lanes)
        if (c.used) HIP_TRY(hipStreamSynchronize(c.st));
    HIP_TRY(hipMemset(s->cells, 0, (size_t)s->cap * sizeof(RoarStreamCell)));
    HIP_TRY(hipMemset(s->dev, 0, dev_bytes(s)));
    HIP_TRY(hipDeviceSynchronize());
    memset(s->pin, 0, pin_bytes(s));
    std::atomic_thread_fence(std::memory_order_seq_cst);
    // the records of the epoch before the one that ends here go (tickets <= epoch_base: asked for again, wait() says so)
    if (s->epoch_base > s->rec_base) {
        const size_t drop = (size_t)(s->epoch_base - s->rec_base);
        for (size_t i = 0; i < drop; ++i)
            if (s->short_of[i] && s->old_short.size() < (1u << 20)) {
                try {
                    s->old_short[s->rec_base + i + 1] = s->short_of[i];
                } catch (...) {
                }
            }
        s->short_of.erase(s->short_of.begin(), s->short_of.begin() + drop);
        s->first_of.erase(s->first_of.begin(), s->first_of.begin() + drop);
        s->done_of.erase(s->done_of.begin(), s->done_of.begin() + drop);
        s->rec_base = s->epoch_base;
    }
    s->epoch_base += s->submitted;
    s->submitted = s->completed = s->claimed = 0;
    s->launched = 0;
    s->max_exited = 0;
    s->pending_q = s->unclaimed_q = 0;
    s->nq_of.clear();
    s->n_resets++;
    return ROAR_OK;
}

}  // namespace

extern "C" {

int roar_search_stream_open(roar_index *idx, uint32_t k, uint32_t L_pq, roar_stream **out) {
    DeviceRestore keep_device_;
    if (!idx || !out) return fail(ROAR_ERR_ARG, "bad arguments");
    *out = nullptr;
    if (!idx->base || idx->n == 0) return fail(ROAR_ERR_ARG, "search before base vectors were loaded");
    if (!idx->adj) return fail(ROAR_ERR_ARG, "search before the projection graph was loaded");
    if (L_pq == 0 || k == 0) return fail(ROAR_ERR_ARG, "k and L_pq must be positive");
    if (k > L_pq) return fail(ROAR_ERR_ARG, "L_pq must greater or equal than k");  // tests/test_search_roargraph.cpp:192-195
    if (idx->visited_mode == ROAR_VISITED_EXACT || idx->visited_mode == ROAR_VISITED_LDS)
        return fail(ROAR_ERR_ARG, "the batch-stream form runs the forgetful on-chip tag table: option visited_mode must be AUTO or LDS_FORGETFUL");
    int e = use_device(idx);
    if (e) return e;
    roar_stream *s = nullptr;
    try {
        s = new roar_stream();
    } catch (...) {
        return fail(ROAR_ERR_DEVICE, "out of host memory");
    }
    s->idx = idx;
    s->k = k;
    s->L = L_pq;
    if (const char *v = getenv("ROAR_STREAM_CAP")) {  // (tests: a small epoch, so that the reset is exercised)
        const long c = atol(v);
        if (c >= 2 && c <= 65536) s->cap = (uint32_t)c;
    }
    while (s->ring > s->cap / 2 && s->ring > 2) s->ring >>= 1;
    auto bail = [&](int code) {
        if (s->pin) (void)hipHostFree(s->pin);
        if (s->cells) (void)hipFree(s->cells);
        if (s->dev) (void)hipFree(s->dev);
        delete s;
        return code;
    };
    if (hipHostMalloc((void **)&s->pin, pin_bytes(s), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer((void **)&s->pin_dev, s->pin, 0) != hipSuccess) {
        (void)hipGetLastError();
        return bail(fail(ROAR_ERR_DEVICE, "search stream: pinned host allocation failed"));
    }
    memset(s->pin, 0, pin_bytes(s));
    s->ctl = reinterpret_cast<RoarStreamCtl *>(s->pin);
    s->slots = reinterpret_cast<RoarStreamSlot *>(s->pin + sizeof(RoarStreamCtl));
    s->done = reinterpret_cast<RoarStreamDone *>(s->pin + sizeof(RoarStreamCtl) + (size_t)s->ring * sizeof(RoarStreamSlot));
    if (roar_dev_malloc((void **)&s->cells, (size_t)s->cap * sizeof(RoarStreamCell)) != hipSuccess ||
        hipMemset(s->cells, 0, (size_t)s->cap * sizeof(RoarStreamCell)) != hipSuccess || roar_dev_malloc((void **)&s->dev, dev_bytes(s)) != hipSuccess ||
        hipMemset(s->dev, 0, dev_bytes(s)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipGetLastError();
        return bail(fail(ROAR_ERR_DEVICE, "search stream: out of device memory"));
    }
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, idx->device) == hipSuccess && khz > 0) s->wall_khz = khz;
    (void)hipGetLastError();
    idx->streams_open.fetch_add(1);
    *out = s;
    return ROAR_OK;
}

int roar_search_stream_submit(roar_stream *s, const float *queries_dev, uint64_t nq, uint32_t qstride_floats, uint32_t *ids_dev, float *dists_dev,
                              uint32_t *cmps_dev, uint32_t *hops_dev, uint32_t *scanned_dev, uint64_t *ticket) {
    DeviceRestore keep_device_;
    if (!s || !ticket) return fail(ROAR_ERR_ARG, "bad arguments");
    *ticket = 0;
    if (nq == 0) return ROAR_OK;  // (ticket 0: nothing to wait for)
    if (!queries_dev || !ids_dev || !dists_dev) return fail(ROAR_ERR_ARG, "bad search arguments");
    roar_index *idx = s->idx;
    if (qstride_floats < idx->dim_pad || (qstride_floats & 3))
        return fail(ROAR_ERR_ARG, "device queries need a stride >= %u floats that is a multiple of 4", idx->dim_pad);
    if (nq >= (1ull << 31)) return fail(ROAR_ERR_ARG, "too many queries in one batch");
    std::unique_lock<std::mutex> lk(s->mu);
    if (s->broken) return fail(ROAR_ERR_DEVICE, "search stream: an earlier call failed; close the stream");
    int e = use_device(idx);
    if (e) return e;
    if (!s->planned) {
        if ((e = plan_stream(s, queries_dev, nq, qstride_floats))) return e;
    } else if (s->provisional) {
        // The plan stands on defaults because no batch so far was a sample one can measure on (resolve_stream_plan).  This batch is, or
        // somebody else's launches have made the calibration meanwhile: the stream is drained ONCE -- every batch complete, every launch
        // over, so that the calibration's launches are timed on an idle chip -- and planned again.  Results do not depend on the plan.
        bool due = nq >= stream_calibration_min_queries(idx);
        if (!due) {
            std::lock_guard<std::mutex> g2(idx->mu);
            const int cap_keep = idx->max_grid;
            idx->max_grid = 0x7FFFFFFF;  // (the key of the largest batches, as resolve_stream_plan makes it)
            due = find_calibration(idx, plan_key(idx, nq, s->L)) != nullptr;
            idx->max_grid = cap_keep;
        }
        if (due) {
            if ((e = wait_for(s, lk, [&] { return s->completed == s->submitted; }, "the drain of the stream"))) return e;
            for (auto &c : s->lanes)
                if (c.used) HIP_TRY(hipStreamSynchronize(c.st));
            if ((e = plan_stream(s, queries_dev, nq, qstride_floats))) return e;
            s->n_replans++;
        }
    }
    if (s->submitted == s->cap && (e = reset_epoch(s, lk))) return e;
    // a slot is free once the batch that held it is complete
    if (s->submitted - s->completed >= s->ring) {
        const uint32_t need = s->submitted - s->ring + 1;
        if ((e = wait_for(s, lk, [&] { return s->completed >= need; }, "the completion of the oldest batch"))) return e;
    }
    const uint32_t b = s->submitted;
    try {
        s->nq_of.push_back((uint32_t)nq);
        s->short_of.push_back(0), s->first_of.push_back(0), s->done_of.push_back(0);
    } catch (...) {
        return fail(ROAR_ERR_DEVICE, "out of host memory");
    }
    volatile RoarStreamSlot *sl = &s->slots[b & (s->ring - 1)];
    sl->seq1 = 0;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    sl->queries = (uint64_t)(uintptr_t)queries_dev;
    sl->out_ids = (uint64_t)(uintptr_t)ids_dev;
    sl->out_dists = (uint64_t)(uintptr_t)dists_dev;
    sl->out_cmps = (uint64_t)(uintptr_t)cmps_dev;
    sl->out_hops = (uint64_t)(uintptr_t)hops_dev;
    sl->out_scanned = (uint64_t)(uintptr_t)scanned_dev;
    sl->nq = (uint32_t)nq;
    sl->qstride = qstride_floats;
    sl->pad = 0;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    sl->seq1 = b + 1u;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    *reinterpret_cast<volatile uint32_t *>(&s->ctl->tail) = b + 1u;  // published
    std::atomic_thread_fence(std::memory_order_seq_cst);             // ... THEN `exited` is read (top_up)
    s->submitted = b + 1;
    s->pending_q += nq;
    s->unclaimed_q += nq;
    poll(s);
    if ((e = top_up(s, false))) return e;
    *ticket = s->epoch_base + b + 1;
    return ROAR_OK;
}

int roar_search_stream_wait(roar_stream *s, uint64_t ticket, uint32_t *short_count) {
    DeviceRestore keep_device_;
    if (!s) return fail(ROAR_ERR_ARG, "bad arguments");
    if (short_count) *short_count = 0;
    if (ticket == 0) return ROAR_OK;
    std::unique_lock<std::mutex> lk(s->mu);
    if (ticket > s->epoch_base + s->submitted) return fail(ROAR_ERR_ARG, "search stream: ticket %llu was never handed out", (unsigned long long)ticket);
    auto answer_old = [&]() {  // (complete when its epoch ended; its record was dropped one epoch later)
        if (short_count) {
            const auto it = s->old_short.find(ticket);
            *short_count = it == s->old_short.end() ? 0u : it->second;
        }
        return ROAR_OK;
    };
    if (ticket <= s->rec_base) return answer_old();
    if (ticket > s->epoch_base) {  // (tickets of earlier epochs were complete when their epoch ended)
        if (s->broken) return fail(ROAR_ERR_DEVICE, "search stream: an earlier call failed; close the stream");
        int e = use_device(s->idx);
        if (e) return e;
        const uint32_t b = (uint32_t)(ticket - s->epoch_base - 1);
        const uint64_t epoch = s->epoch_base;
        if ((e = wait_for(s, lk, [&] { return s->epoch_base != epoch || batch_done(s, b); }, "the completion of the batch"))) return e;
        if (s->epoch_base == epoch && b >= s->completed) {  // finished before an earlier batch: read its record directly
            volatile RoarStreamDone *d = &s->done[b & (s->ring - 1)];
            std::atomic_thread_fence(std::memory_order_acquire);
            s->short_of[ticket - 1 - s->rec_base] = d->nshort, s->first_of[ticket - 1 - s->rec_base] = d->t_first, s->done_of[ticket - 1 - s->rec_base] = d->t_done;
        }
    }
    if (ticket <= s->rec_base) return answer_old();  // (two epochs went by during the wait)
    if (short_count) *short_count = s->short_of[ticket - 1 - s->rec_base];
    return ROAR_OK;
}

int roar_search_stream_times(roar_stream *s, uint64_t ticket, uint64_t *first_ns, uint64_t *done_ns) {
    if (!s || ticket == 0) return fail(ROAR_ERR_ARG, "bad arguments");
    std::lock_guard<std::mutex> guard(s->mu);
    if (ticket > s->epoch_base + s->submitted) return fail(ROAR_ERR_ARG, "search stream: ticket %llu was never handed out", (unsigned long long)ticket);
    if (ticket <= s->rec_base)
        return fail(ROAR_ERR_ARG, "search stream: the times of ticket %llu are no longer kept (they are for the running epoch of %u batches and the one before it)",
                    (unsigned long long)ticket, s->cap);
    if (ticket > s->epoch_base) {
        poll(s);
        const uint32_t b = (uint32_t)(ticket - s->epoch_base - 1);
        if (!batch_done(s, b)) return fail(ROAR_ERR_ARG, "search stream: the batch is not complete yet (wait for it first)");
        if (b >= s->completed) {
            volatile RoarStreamDone *d = &s->done[b & (s->ring - 1)];
            s->first_of[ticket - 1 - s->rec_base] = d->t_first, s->done_of[ticket - 1 - s->rec_base] = d->t_done;
        }
    }
    const double ns_per_tick = 1.0e6 / (double)s->wall_khz;
    if (first_ns) *first_ns = (uint64_t)((double)s->first_of[ticket - 1 - s->rec_base] * ns_per_tick);
    if (done_ns) *done_ns = (uint64_t)((double)s->done_of[ticket - 1 - s->rec_base] * ns_per_tick);
    return ROAR_OK;
}

int roar_search_stream_info(roar_stream *s, const char *key, int64_t *value) {
    if (!s || !key || !value) return fail(ROAR_ERR_ARG, "bad arguments");
    std::lock_guard<std::mutex> guard(s->mu);
    const std::string k = key;
    if (k == "launches") *value = s->n_launches;
    else if (k == "workgroups_launched") *value = (int64_t)s->launched;
    else if (k == "workgroups_alive") *value = s->planned ? (int64_t)alive(s) : 0;
    else if (k == "resident_capacity") *value = s->R;
    else if (k == "heals") *value = s->n_heals;
    else if (k == "debug_break") {  // tests: the stream as a failed launch or a timed-out wait leaves it
        s->broken = true;
        *value = 1;
    } else if (k == "epochs") *value = s->n_resets;
    else if (k == "replans") *value = s->n_replans;
    else if (k == "plan_provisional") *value = s->provisional ? 1 : 0;
    else if (k == "batches_submitted") *value = (int64_t)(s->epoch_base + s->submitted);
    else if (k == "batches_completed") {
        poll(s);
        *value = (int64_t)(s->epoch_base + s->completed);
    } else if (k == "ring") *value = s->ring;
    else if (k == "lds_bytes") *value = s->planned ? (int64_t)s->plan.p.lds_bytes : 0;
    else if (k == "screen") *value = s->planned && s->plan.p.screen ? 1 : 0;
    else if (k == "vis_entries") *value = s->planned ? (int64_t)(((s->plan.p.vis_sets_log2 >> 31) ? 12u : ((s->plan.p.vis_sets_log2 >> 28) & 1u) ? 20u : 4u) << (s->plan.p.vis_sets_log2 & 31u)) : 0;
    else if (k == "tag_protect") *value = s->planned ? s->plan.sel.protect : 0;
    else if (k == "register_pool") *value = s->planned ? s->plan.sel.regpool : 0;
    else if (k == "wall_clock_khz") *value = s->wall_khz;
    else if (k == "take_ns_total" || k == "take_calls" || k == "busy_ns_total") {  // what the workgroups' dequeues / searches cost (device clock, all workgroups together, this epoch)
        unsigned long long v[3] = {0, 0, 0};
        if (s->dev && hipMemcpy(v, &s->dev->take_ticks, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return fail(ROAR_ERR_DEVICE, "search stream: cannot read the counters");
        *value = k == "take_calls" ? (int64_t)v[1] : (int64_t)((double)(k == "busy_ns_total" ? v[2] : v[0]) * 1.0e6 / (double)s->wall_khz);
    }
    else return fail(ROAR_ERR_ARG, "Invalid parameter name.");
    return ROAR_OK;
}

int roar_search_stream_close(roar_stream *s) {
    DeviceRestore keep_device_;
    if (!s) return ROAR_OK;
    int rc = ROAR_OK;
    bool launches_over = true;  // every launch of the stream is known to have ended (cleanly or not): nothing reads its buffers or the index
    {
        std::unique_lock<std::mutex> lk(s->mu);
        (void)use_device(s->idx);
        if (!s->broken && s->planned) rc = wait_for(s, lk, [&] { return s->completed == s->submitted; }, "the drain of the stream");
        if (rc == ROAR_OK && !s->broken) {
            // every batch is complete: the workgroups find nothing published and leave; their launches end
            for (auto &c : s->lanes)
                if (c.used && hipStreamSynchronize(c.st) != hipSuccess) rc = fail(ROAR_ERR_DEVICE, "search stream: a launch failed: %s", hipGetErrorString(hipGetLastError()));
        } else {
            // A stream that failed or did not drain: its workgroups are bounded on the device (one that waits for another gives up after
            // STREAM_SPIN_LIMIT_TICKS = 2 s, the others finish the queries they hold), so their launches are given CLOSE_GRACE_S to end.
            // Only then may anything they read be freed -- and only then is the index let go (roar_destroy / roar_set_base / a new graph
            // would otherwise free base rows, adjacency and screening copy under running workgroups).
            const std::string keep = g_err;
            const auto t0 = std::chrono::steady_clock::now();
            for (auto &c : s->lanes) {
                if (!c.used) continue;
                for (;;) {
                    const hipError_t q = hipEventQuery(c.ev);
                    if (q != hipErrorNotReady) break;  // (ended; an error status is an ended launch too)
                    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > CLOSE_GRACE_S) {
                        launches_over = false;
                        break;
                    }
                    lk.unlock();
                    std::this_thread::sleep_for(std::chrono::milliseconds(1));
                    lk.lock();
                }
                if (!launches_over) break;
            }
            (void)hipGetLastError();
            g_err = keep;
            if (rc == ROAR_OK)
                rc = fail(ROAR_ERR_DEVICE, launches_over ? "search stream: an earlier call failed; the stream is closed, its results are not to be used"
                                                         : "search stream: an earlier call failed and its workgroups are still running: the stream's buffers are "
                                                           "not freed and the index stays pinned (roar_destroy will refuse it)");
        }
    }
    if (!launches_over) return rc;  // (nothing freed, streams_open kept: the index must outlive workgroups that may still read it)
    s->idx->streams_open.fetch_sub(1);
    for (auto &c : s->lanes) {
        (void)hipEventDestroy(c.ev);
        (void)hipStreamDestroy(c.st);
    }
    (void)hipFree(s->cells);
    (void)hipFree(s->dev);
    (void)hipHostFree(s->pin);
    delete s;
    return rc;
}

}  // extern "C"
