// roar_device.h -- launch-parameter block shared by the HIP kernels (roar_kernels.hip) and the
// C-ABI host code (roar_capi.cpp).  Internal; the public boundary is include/roargraph_amd.h.
#pragma once
#include <stdint.h>

// ---- batch-stream form (roar_search_stream_*, roar_stream.cpp; kernel: SRC == 2 in roar_search_v2.h) -----------------------------
// Workgroups do not belong to a launch's batch: they take queries from a ring of SUBMITTED batches, cross batch boundaries, and
// leave when nothing published is left.  The host publishes a batch by writing its slot and then `tail` in PINNED HOST memory; the
// kernel reports a finished batch in the slot's RoarStreamDone (also pinned).  A read of host memory is a PCIe round trip, and the
// chip can only keep so many of them in flight (first form: every workgroup fetched every batch's slot itself -- 6 144 reads per
// batch, 0.7 ms per batch whatever its size), so the workgroups keep what they have learnt in DEVICE memory (RoarStreamDev): ONE
// workgroup fetches a batch's slot (a compare-and-swap on the batch's state word elects it) and leaves a copy for the others, and the
// host's `tail` is looked at by one workgroup at a time, at most every few microseconds, and mirrored.  The per-batch counters
// (RoarStreamCell) are indexed by batch NUMBER and never recycled while any workgroup is alive, so a late workgroup can touch the
// counters of a long finished batch without harm.
struct RoarStreamSlot {   // 64 bytes, ring position = batch number % ring; written by the host (seqlock: seq1 = 0 first, the fields, seq1 last)
    uint64_t queries;     // device pointers of the batch (out_cmps / out_hops / out_scanned may be 0)
    uint64_t out_ids, out_dists, out_cmps, out_hops, out_scanned;
    uint32_t nq;
    uint32_t qstride;     // floats
    uint32_t pad;
    uint32_t seq1;        // batch number + 1 (0 while the slot is being rewritten)
};
struct RoarStreamDone {   // 32 bytes, one per ring position; pinned host memory, written by the kernel
    uint64_t t_first;     // device wall clock (100 MHz) when the batch's first query was taken ...
    uint64_t t_done;      // ... and when its last query had its results in device memory
    uint32_t nshort;      // queries that returned fewer than k results
    uint32_t seq1_done;   // batch number + 1 once every query of the batch has its results in device memory (written last)
    uint32_t seq1_claimed;  // batch number + 1 once the batch's last query has been handed out (the host then starts no workgroups for it)
    uint32_t pad;
};
struct RoarStreamCtl {    // 64 bytes, pinned host memory
    uint32_t tail;        // batches published so far (host writes; a workgroup reads it when the device's mirror says "nothing left")
    uint32_t exited;      // workgroups that have left: posted by every 64th one and by the last one alive (the host keeps the largest value seen)
    uint32_t error;       // != 0: a workgroup gave up waiting for another one (2 s of device clock; never seen outside fault injection) -- the
                          // host fails the stream instead of waiting for a kernel that spins: 1 slot fetch, 2 batch count, 3 tail look
    uint32_t pad[13];
};
struct RoarStreamCell {   // device memory, one per batch number
    unsigned long long done_short;  // low word: queries finished; high word: of those, short ones (one 64-bit add per query)
    uint32_t next;                  // next query of the batch to hand out (grows past nq: harmless)
    uint32_t pad;                   // (NOTHING that is merely LOADED may live in these lines: see RoarStreamDev::state)
};
struct RoarStreamDev {    // device memory: what the workgroups have learnt from the host, shared among them.  One 128-byte line per word
                          // that many workgroups touch at once (atomics and agent-scope loads on one line are served one after the other:
                          // first form, everything in one line -- the last batch of a stream finished 3 ms late behind 3 000 leaving workgroups)
    unsigned long long tail_time;   // 40 bits of the device wall clock when the last look at the host's tail came back << 24 | that tail
    unsigned long long pad0[15];
    unsigned long long started_exited;  // workgroups started << 32 | workgroups that have left (one 64-bit add each way)
    unsigned long long pad1[15];
    uint32_t checking;    // 1 while a workgroup is reading the host's tail (the others wait for its answer instead of asking themselves)
    uint32_t pad2[31];
    uint32_t frontier;    // every batch before this one has handed out all its queries (set by whoever takes a batch's last query)
    uint32_t pad3[31];
    unsigned long long take_ticks;  // device clock ticks workgroups have spent in stream_next (count + take), all of them together ...
    unsigned long long take_calls;  // ... over this many calls (info keys "take_ns_total" / "take_calls": what a dequeue costs)
    unsigned long long busy_ticks;  // ... and between a take and the end of its query's search (info key "busy_ns_total")
    unsigned long long pad4[13];
    // Per batch number: its entry of `mirror` is 0 not fetched, 1 being fetched, 2 there, 3 the host's slot already held a later batch.
    // Its own array, NOT a word of the batch's counter cell: a line that takes millions of atomic adds a second must not also be
    // LOADED by every workgroup -- each load puts a copy of the line into an XCD's L2 that the next add has to chase out again, and
    // that traffic slows every memory access of the chip, not the loads (measured: one load of the cell's line per query made every
    // hop of every search 20-30 % slower, 9.5 -> 7.6 M QPS at L_pq = 30; extra adds on lines nobody loads cost nothing).
    uint32_t state[65536];
    RoarStreamSlot mirror[1];       // [cap] copies of the host's slots by batch NUMBER, each written once, by the workgroup that fetched it (state 2: there for good)
};

struct RoarSearchParams {
    // base vectors: f32 rows, `ld` floats apart, scored over `dim` floats (dim % 8 == 0, zero padded)
    const float *base;
    uint64_t ld;
    uint32_t dim;
    // SearchRoarGraph kernels, 128-d / 200-d (inner product) / 512-d rows: the 8-bit screening copy of the base rows,
    // roar_screen8_stride(dim) bytes per row (dim codes, scale, bounds; roar_launch_make_screen8, roar_search_v2.h
    // screen8i_pass / screen8_pass); null = off.  With it the LDS block at off_q holds the query's own 8-bit codes (dim
    // bytes) behind its dim floats
    const unsigned char *screen;
    // inner product: the centre of the screening copy's codes (they code x - mu; dim floats), null = not centred
    const float *screen_mu;
    // graph, fixed-stride adjacency rows: word 0 = degree, word 1 + j = j-th out-neighbour
    const uint32_t *adj;
    uint32_t adj_stride;
    // "edge floats" (sel.ann; inner product, rows whose screening codes fill whole lines: 128-d, 512-d): a second array parallel to the
    // adjacency, 8 bytes per adjacency WORD -- entry w of row v = the scale and the two bf16 bounds of the screening row of v's
    // neighbour in word w (roar_screen8i_pass) -- fetched with the adjacency row, in the same round trip, so that a screened
    // neighbour costs its code row's whole lines and nothing else (the 16 bytes of floats were a line request of their own wherever
    // they lived alone).  LDS: 8 bytes per candidate of a round trip at off_cmeta.
    const unsigned long long *adj_meta;
    uint32_t off_cmeta;
    uint32_t n;
    uint32_t ep;
    // query batch
    const float *queries;
    uint64_t qstride;
    uint32_t nq;
    uint32_t k;
    uint32_t L;
    // outputs (cmps / hops / scanned / short_count may be null)
    uint32_t *out_ids;
    float *out_dists;
    uint32_t *out_cmps;
    uint32_t *out_hops;
    uint32_t *out_scanned;
    uint32_t *short_count;
    // launch cell, u32[4] = {next query to dequeue, workgroups that have left, short queries, -}.  SearchRoarGraph
    // kernels: zero on entry, zeroed again by the last workgroup to leave, which also stores the short count to *short_count and
    // the launch's own stamp -- its use of the cell, counted from 1 -- to *stamp (a word of its own, NOT in the cell: every workgroup
    // loads it once, and a line that takes an atomic add per query must not be loaded, RoarStreamDev::state).  A launch that finds
    // another stamp than its predecessor's (cell_prev) does not search: the launch that used the cell before it did not finish (a
    // device fault, or more launches in flight than the ring has cells) -- it stores 0xFFFFFFFF to *short_count and leaves.
    // Two-hop kernels: word 0 only, zeroed by the host before every launch.
    uint32_t *counter;
    uint32_t *stamp;
    uint32_t cell_prev, cell_stamp;
    // visited, LDS mode: sets (power of two) x 4 ways of u16 tags
    uint32_t vis_sets_log2;  // search kernel only: bit 31: the table has 3 << (value & 31) sets (bit 28: 5 << ...); bit 30: forgetful form (a
                             // full set overwrites a victim; else ids that find their set full go to the spill list below);
                             // bit 29 (forgetful form, LDS pool): bit 15 of a tag = "its node is in the pool", never the victim
    // host-buffer form with a pageable query block: rows [0, *ready) of `queries` (pinned staging) have been written by the
    // host thread that is still packing the rest; null = every row is there (search kernel only)
    const uint32_t *ready;
    // visited, EXACT mode: one u8[tag_stride] list per resident workgroup + its epoch.  LDS mode, exact form: the spill
    // lists, one u16[tag_stride] list per resident workgroup (tag_stride in elements) + its epoch
    uint8_t *tags;
    uint64_t tag_stride;
    uint32_t *epochs;
    // LDS carve-up (byte offsets, all 16-byte aligned)
    uint32_t off_pool, off_ck, off_cid, off_q, off_vis, off_ctrl, lds_bytes;
    // bipartite two-hop walk only (roar_bipartite_kernel): CSR over base + training-query nodes,
    // explicit entry points [nq][n_init], 768 B of LDS at off_bp
    const uint64_t *bp_off;
    const uint32_t *bp_nbrs;
    const uint32_t *init_ids;
    uint32_t n_init;
    uint32_t off_bp;
    // eight-waves-per-query form: one u32[tag_stride32] epoch list per resident workgroup (test-and-set)
    uint32_t *tags32;
    uint64_t tag_stride32;
    // hub-annotated copy of the adjacency (long pools): rows of hub_stride words, word 0 = degree | hubs << 16, hub
    // neighbours first, one u16 bitmap slot per hub neighbour behind the ids; LDS: 256-word row buffer + 8 KB bitmap
    const uint32_t *adj_hub;
    uint32_t hub_stride;
    uint32_t off_rowbuf, off_hbits;
    // per-query phase timers of the instrumented kernels (u64[nq][8]: total, select, adjacency, filter, score, merge
    // shader-clock cycles of wave 0, speculative adjacency fetches that were used, adjacency round trips paid); may be null
    unsigned long long *prof;
    // SearchRoarGraph kernels: what each query asked the memory side for (roar_set_traffic_counters_device; may be null),
    // u32[nq][4] = {rows scored on the 8-bit screening copy, rows scored on the f32 rows (entry point included), adjacency
    // words fetched (whole 64-word chunks; speculative fetches that were dropped included), 0}
    uint32_t *out_traffic;
    // batch-stream form (sel.stream): device addresses of the pinned control blocks, the per-batch counters, ring size (a power of
    // two), the first batch number a workgroup of THIS launch looks at (every earlier batch is complete), and the LDS block (128 bytes)
    // that holds a workgroup's copy of the slot it works on
    const RoarStreamCtl *s_ctl;
    const RoarStreamSlot *s_slots;
    RoarStreamDone *s_done;
    RoarStreamCell *s_cells;
    RoarStreamDev *s_dev;
    uint32_t s_ring;
    uint32_t s_start;
    uint32_t s_tail;  // batches published when this launch was made
    uint32_t off_stream;
};

// Which kernel a launch runs.
enum { ROAR_FORM_SEARCH = 0, ROAR_FORM_BIPARTITE = 2, ROAR_FORM_BIPARTITE_WG = 3, ROAR_FORM_BUILDER = 4 };
struct RoarKernelSel {
    int metric_ip;     // 1: negated inner product, 0: squared L2
    int visited_mode;  // 0 exact tags in HBM, 1 LDS tag table
    int form;          // ROAR_FORM_*
    int waves;         // SEARCH: waves per query (1, 2, 4)
    int regpool;       // SEARCH / BUILDER: candidate pool in a register pair of wave 0 (L_pq <= 64) instead of LDS
    int prof;          // SEARCH: instrumented instantiation (phase timers; Text-to-Image shape only)
    int hub;           // SEARCH: hub-bitmap form (LDS pool only)
    int gated;         // SEARCH, one wave per query, no hub / prof: queries become readable while the launch runs (RoarSearchParams::ready)
    int stats;         // SEARCH, one wave per query, no hub / prof / gate: the instantiation that fills RoarSearchParams::out_traffic
    int protect;       // SEARCH, LDS visited mode, one wave per query, pool in LDS, no hub / gate: forgetful table that keeps pool members' tags
    int ann;           // SEARCH, LDS visited mode, one wave per query, no hub / prof / stats / gate: the neighbours' screening floats come with the adjacency row (RoarSearchParams::adj_meta)
    int stream;        // SEARCH, LDS visited mode (forgetful), one wave per query, no hub / prof / gate / stats: queries come from a ring of submitted batches
};

struct RoarDistParams {
    const float *base;
    uint64_t ld;
    uint32_t dim;
    const float *queries;
    uint64_t qstride;
    uint32_t nq;
    const uint32_t *ids;  // [nq][m]
    uint32_t m;
    float *out;           // [nq][m]
    uint32_t off_ck, off_cid, off_q, lds_bytes;
};

// The screening copy.  A row of codes is dim / 16 chunks of 16 codes; its 16 bytes of floats (inner product: the last dim % 16 (0 or 8)
// codes, the scale and the two bounds; L2: scale, lo, |x~|^2, bound) live
//   * INSIDE the row where the codes do not fill whole 128-byte lines anyway (200-d: 208 -> 256 bytes, two lines instead of two or three),
//   * in a packed array BEHIND the code rows (16 bytes per row, at byte n x stride of the copy) where they do (128-d, 512-d): round 4
//     kept them in the row -- 528 bytes, always five lines for 4.1 lines of payload at 512-d; now four whole lines + one 16-byte
//     read (eight rows' floats share a line, and hub rows' lines stay in L2).
static inline constexpr bool roar_screen8_split(uint32_t dim) { return dim % 128u == 0u; }
static inline constexpr uint32_t roar_screen8_stride(uint32_t dim) {
    return roar_screen8_split(dim) ? dim
           : (112u + (dim / 16u + 1u) * 16u + 127u) / 128u <= ((dim / 16u + 1u) * 16u + 127u) / 128u ? (dim / 16u + 1u) * 16u
                                                                                                     : ((dim / 16u + 1u) * 16u + 127u) & ~127u;
}
// bytes a scored row asks the memory side for / bytes of the whole copy of n rows
static inline constexpr uint32_t roar_screen8_row_bytes(uint32_t dim) { return roar_screen8_stride(dim) + (roar_screen8_split(dim) ? 16u : 0u); }
static inline constexpr uint64_t roar_screen8_bytes(uint32_t dim, uint64_t n) { return n * (uint64_t)roar_screen8_row_bytes(dim); }
// L2 form: allowance for the rounding of |x~|^2 - 2 x~.q + |q|^2 as the kernel evaluates it, in units of |x~|^2 + |q|^2
static inline constexpr double roar_screen8_l2_rounding(uint32_t dim) {
    double y = (double)dim;
    for (int i = 0; i < 40; ++i) y = 0.5 * (y + (double)dim / y);
    return 1.25 * (2.0 * dim * (1.0 + y) + dim + 8.0) / 16777216.0 * 1.0001;
}

#ifdef __cplusplus
extern "C" {
#endif
// Host-side launchers implemented in roar_kernels.hip.  `stream` is a hipStream_t.  Return a hipError_t value (0 = success).
int roar_launch_search(const RoarSearchParams *p, const RoarKernelSel *sel, uint32_t grid, void *stream);
int roar_launch_distances(const RoarDistParams *p, int metric_ip, void *stream);
int roar_launch_merge_topk(uint32_t nshards, uint64_t nq, uint32_t k, const uint32_t *ids, const float *dists,
                           uint32_t *out_ids, float *out_dists, void *stream);
// f32 rows of dim floats (dim % 8 == 0, dim <= 1024) at stride ld -> screening rows (see RoarSearchParams::screen);
// out: roar_screen8_bytes(dim, n) bytes (code rows, then -- roar_screen8_split -- the rows' floats)
// l2 != 0: the L2 form of the row (dim % 16 == 0): dim codes u_i with x~_i = lo + s u_i, then s, lo, |x~|^2, bound
// mu (inner-product form only; may be null): dim floats the codes are centred on
int roar_launch_make_screen8(const float *base, uint64_t ld, uint32_t dim, uint64_t n, unsigned char *out, int l2, const float *mu, void *stream);  // out: roar_screen8_bytes(dim, n)
int roar_launch_rebase_ids(uint32_t *ids, uint64_t count, uint32_t base, void *stream);
// RoarSearchParams::adj_meta from the adjacency rows and the screening copy's packed floats (16 bytes per base row: .z scale, .w bounds)
int roar_launch_edge_floats(const uint32_t *adj, uint32_t stride, uint64_t n, const void *row_floats, unsigned long long *out, void *stream);
// tests: the search's own screening passes over explicit (query, row) pairs, every row against its own threshold (roar_search_v2.h,
// roar_screen_check_kernel); _1 = inner product, _0 = L2
int roar_launch_screen_check_1(const unsigned char *scr, const float *mu, uint32_t dim, const float *queries, uint64_t qstride, uint64_t nq,
                               const uint32_t *ids, float *chk, uint32_t m, uint64_t n_rows, void *stream);
int roar_launch_screen_check_0(const unsigned char *scr, const float *mu, uint32_t dim, const float *queries, uint64_t qstride, uint64_t nq,
                               const uint32_t *ids, float *chk, uint32_t m, uint64_t n_rows, void *stream);
// Max resident workgroups per CU for the selected kernel at this LDS size.
int roar_search_occupancy(const RoarKernelSel *sel, uint32_t dim, uint32_t lds_bytes, int *blocks_per_cu);
#ifdef __cplusplus
}
#endif
