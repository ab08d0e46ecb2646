/*
 * roargraph_amd.h -- C ABI of the MI355X-native RoarGraph search engine.
 *
 * This is the drop-in boundary for ONE path of matchyc/mysteryann: the batched best-first beam
 * search over the projected (RoarGraph) graph, i.e. IndexBipartite::SearchRoarGraph and what it
 * calls.  Everything below the line is hand-written HIP for gfx950; there is no CPU fallback --
 * every compute entry point fails with ROAR_ERR_DEVICE when no HIP device is present.
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference
 * tree).  Plain pointers and sizes only; no C++ or torch types.  All functions return ROAR_OK (0)
 * or a negative code and never throw; roar_last_error() returns the thread-local message, which
 * for the cases the reference throws on carries the reference's own text.
 */
#ifndef ROARGRAPH_AMD_H
#define ROARGRAPH_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct roar_index roar_index;

/* include/efanna2e/distance.h:15 (FAST_L2 = 2 and PQ = 3 are never instantiated on the path) */
enum { ROAR_METRIC_L2 = 0, ROAR_METRIC_IP = 1, ROAR_METRIC_COSINE = 4 };

enum {
    ROAR_OK = 0,
    ROAR_ERR_ARG = -1,    /* bad argument / state (std::invalid_argument in the reference) */
    ROAR_ERR_IO = -2,     /* cannot open file (reference: exit(-1) util.h:109-112 / "cannot open file") */
    ROAR_ERR_FORMAT = -3, /* "Data file size wrong!" (util.h:124,205-208) */
    ROAR_ERR_DEVICE = -4, /* no HIP device, HIP runtime error, out of device memory */
    ROAR_ERR_SHORT = -5   /* "not enough results: x, expected: k" (src/index_bipartite.cpp:2408-2412) */
};

/* visited-set policy, option "visited_mode" (replaces VisitedList / VisitedListPool,
 * include/visited_list_pool.h:8-74):
 *   EXACT          per-workgroup u8 epoch tags in HBM: mass[id]==curV semantics.
 *   LDS            per-query set-associative tag table in LDS, exact: the table never gives an entry up, an id that
 *                  finds its set full is looked up / marked in the workgroup's spill list in HBM (u16 epoch tags).
 *   LDS_FORGETFUL  the same table, a full set overwrites a victim entry (never a false positive): a forgotten node
 *                  is scored again if met again; nothing is ever stored to HBM.
 * ids / distances / hops are identical to the reference in every mode; cmps is the reference's in EXACT and LDS and can
 * only be >= it in LDS_FORGETFUL.  (If HBM cannot hold one spill list per resident workgroup -- 2 n bytes each -- an LDS
 * launch runs forgetful; info key "last_visited_forgetful" says so.) */
enum { ROAR_VISITED_EXACT = 0, ROAR_VISITED_LDS = 1, ROAR_VISITED_AUTO = 2 /* default: the fastest form for the launch, measured on the loaded index (option "calibrate") -- LDS_FORGETFUL for short and medium pools, EXACT for the longest */, ROAR_VISITED_LDS_FORGETFUL = 3 };

/* ---- lifecycle ------------------------------------------------------------------------------
 * IndexBipartite::IndexBipartite(dimension, n, metric, nullptr)  src/index_bipartite.cpp:30-38,
 * Index::Index metric -> distance choice                          src/index.cpp:8-25.
 * `dim` is the UNPADDED vector dimension; rows are padded to a multiple of 8 floats on load
 * (util.h:37-75).  `device` is the HIP device ordinal. */
int roar_create(roar_index **out, uint32_t dim, int metric, int device);
void roar_destroy(roar_index *idx);
const char *roar_last_error(void);

/* ---- base vectors ---------------------------------------------------------------------------
 * IndexBipartite::LoadVectorData / LoadSearchNeededData  src/index_bipartite.cpp:2661-2692,
 * load_meta / load_data / data_align / normalize         include/efanna2e/util.h:37-75,106-127,179-225.
 * COSINE indexes L2-normalise every row before upload (src/index_bipartite.cpp:2676-2681). */
int roar_load_base_fbin(roar_index *idx, const char *path);
/* Same, from caller memory: n rows of `dim` valid floats at `row_stride_floats` stride (host). */
int roar_set_base(roar_index *idx, const float *rows, uint64_t n, uint32_t row_stride_floats);
/* Adopt rows already resident in HBM (no copy, caller keeps ownership and must outlive the index).
 * row_stride_floats must be a multiple of 4 and >= the padded dim with zero padding in place.
 * The rows must not change while the index uses them: derived copies (the 8-bit screening copy, option "screen") are made
 * from them in this call; call it again after changing them. */
int roar_set_base_device(roar_index *idx, const float *rows_dev, uint64_t n, uint32_t row_stride_floats);

/* ---- graph ----------------------------------------------------------------------------------
 * IndexBipartite::LoadProjectionGraph  src/index_bipartite.cpp:2097-2117
 * IndexBipartite::SaveProjectionGraph  src/index_bipartite.cpp:2606-2619
 * File: u32 ep; u32 npts; npts x {u32 deg; u32 nbr[deg]}.  Neighbour order is preserved (it is
 * the scan order of the search).  The CSR form is projection_graph_ (index_bipartite.h:150). */
int roar_load_graph_file(roar_index *idx, const char *path);
int roar_set_graph_csr(roar_index *idx, uint32_t ep, uint64_t n, const uint64_t *offsets, const uint32_t *nbrs);
int roar_save_graph_file(roar_index *idx, const char *path);
/* Copy the graph back out (host buffers; pass NULL nbrs to query sizes only). */
int roar_get_graph_csr(roar_index *idx, uint32_t *ep, uint64_t *n, uint64_t *nedges, uint64_t *offsets,
                       uint32_t *nbrs);

/* ---- bootstrap builder (scaffolding for measurement, NOT the reference's builder) ----------------
 * No index can be downloaded to the GPU box, so one is built in HBM from the loaded base rows, with
 * the search kernel as the work-horse:
 *   phase 1  incremental batch insertion: each node is searched in the graph built so far (L_build),
 *            gets its M nearest as out-edges and is appended to their rows while they hold fewer than
 *            base_cap entries; entry point = node nearest the centroid (CalculateProjectionep,
 *            src/index_bipartite.cpp:2004-2041).
 *   phase 2  projection of n_learn training queries (resident in HBM, drawn from the QUERY
 *            distribution), `rounds` times: search (L_learn) -> M_sq nearest base points; pivot = the
 *            nearest; the others are occlusion-pruned against the pivot to at most M_pj and linked
 *            pivot <-> kept.  This follows the shape of BuildRoarGraph / LinkProjection /
 *            PruneBiSearchBaseGetBase (src/index_bipartite.cpp:143-233, 1043-1093, 1612-1694) but uses
 *            approximate neighbour lists from the graph itself instead of an exact learn->base kNN file.
 * Rows never exceed max_degree.  The result is an ordinary projection graph: roar_save_graph_file
 * writes the reference's index format.  The reference's own builder stays SURVEY 8(f) "next" #1. */
typedef struct roar_build_params {
    uint32_t M;          /* out-edges of a new node in phase 1 (<= 64) */
    uint32_t base_cap;   /* row capacity during phase 1 */
    uint32_t max_degree; /* final row capacity */
    uint32_t L_build;    /* pool length of the phase-1 searches */
    uint64_t max_batch;  /* nodes / queries per launch (0 = 2^19) */
    const float *learn_dev; /* training queries in HBM, may be NULL */
    uint64_t n_learn;
    uint32_t learn_stride_floats;
    uint32_t M_sq;       /* neighbours taken per training query (<= 64); run_roargraph_test.sh:9 uses 100 */
    uint32_t M_pj;       /* kept per pivot after pruning; the reference's M_pjbp (35) */
    uint32_t L_learn;    /* pool length of the phase-2 searches */
    uint32_t rounds;     /* passes over the training queries */
    uint32_t phase1_l2;  /* 1: phase 1 links base points by L2 whatever the index metric is */
    uint64_t n_knn;      /* rows of learn_knn_dev */
    const uint32_t *learn_knn_dev; /* optional exact learn->base kNN ids in HBM, [n_knn][M_sq], nearest first:
                                      the input of the reference builder (LoadLearnBaseKNN,
                                      src/index_bipartite.cpp:2622-2639); projected before the search rounds */
    uint32_t supply_k;   /* phase 3: link every node with its supply_k nearest (index metric) found by
                            searching the finished graph with the node's own vector; 0 = skip */
    uint32_t supply_L;
    uint32_t L_learn_last; /* pool length of the last projection round (0 = L_learn) */
    uint32_t phase1_cand;  /* > 0: phase 1 takes this many search results per node (<= 63) and occlusion-prunes
                              them to M instead of linking the M nearest as they come */
    uint32_t supply_cand;  /* > 0: phase 3 occlusion-prunes this many search results (<= 63) to supply_k */
    uint32_t proj_mode;    /* projection pruning: 0 occlusion under the index metric (reference rule), 1 occlusion
                              under L2, 2 no occlusion: the M_pj nearest to the training query */
    uint32_t supply_mode;  /* same choices for the supply pass */
    uint32_t no_skeleton;  /* 1: the reference's shape -- no phase 1; exact lists (learn_knn_dev) are projected with
                              rows capped at M_pj, then the supply pass links every node, in growing batches */
    uint32_t trim_min_proj;  /* at the end: rows with >= trim_min_proj projection / supply edges keep only their */
    uint32_t trim_keep_base; /* first trim_keep_base phase-1 edges (0 / 0 = keep everything).  Applied only when
                                phase 1 used another metric than the index (phase1_l2 on an IP / COSINE index) */
    uint32_t row_keep;       /* 0: rows are filled first come first kept up to max_degree.  > 0: max_degree is only the
                                room in which a row collects proposals; after every phase a row longer than row_keep
                                keeps its row_keep neighbours nearest to the node itself (index metric), nearest first */
} roar_build_params;
int roar_build_graph(roar_index *idx, const roar_build_params *params);

/* ---- BuildRoarGraph ---------------------------------------------------------------
 * IndexBipartite::LoadLearnBaseKNN + BuildRoarGraph  src/index_bipartite.cpp:2622-2639, 143-233; CLI
 * tests/test_build_roargraph.cpp:105-137.  `learn_knn`: [n_learn][K] nearest base ids of every training query,
 * nearest first (the "learn base knn" file: u32 n; u32 K; ids; host array, or HBM for the _device form -- e.g. straight
 * from roar_exact_knn_device).  The steps of LinkProjection (:1043-1277) in a batch-synchronous form (what maps to
 * what: csrc/roar_rgbuild.hip): entry point (:2004-2041); projection -- pivot = list[0], the next M_sq - 1 (<= 254)
 * candidates pruned by PruneBiSearchBaseGetBase (:1612-1694) to M_pjbp, the pivot's row = the list of its last training
 * query, reverse edges with the re-prune of a full row (:1391-1432, :1527-1610); connectivity enhancement -- every node
 * searched in the supply graph (L_pjpq), the WHOLE final pool of that search pruned by
 * PruneProjectionBaseSearchCandidates (:1846-1940), reverse supply edges with cap 2 M_pjbp (:1352-1389, :1434-1525),
 * nodes walked in growing batches instead of one by one; merge of up to 2 M_pjbp supply edges (:1262-1275).
 * Deterministic (the reference is not: OpenMP + per-node locks), so parity with it is degree statistics + the
 * recall-vs-distance-evaluations curve (tests/test_gpu_parity.py against tests/golden/builder_yardstick_200000.json, the
 * restated reference builder's curve), not bytes. */
int roar_build_roargraph(roar_index *idx, const uint32_t *learn_knn, uint64_t n_learn, uint32_t K, uint32_t M_sq,
                         uint32_t M_pjbp, uint32_t L_pjpq);
int roar_build_roargraph_device(roar_index *idx, const uint32_t *learn_knn_dev, uint64_t n_learn, uint32_t K, uint32_t M_sq,
                                uint32_t M_pjbp, uint32_t L_pjpq);

/* ---- search ---------------------------------------------------------------------------------
 * IndexBipartite::SearchRoarGraph  src/index_bipartite.cpp:2311-2420, batched over nq queries
 * (the reference's `#pragma omp parallel for` over queries, tests/test_search_roargraph.cpp:203-209,
 * becomes one launch).  Per query: ids[k], dists[k] (IP: negated dot, L2: squared), cmps, hops
 * (the pair the reference returns, :2419) and optionally scanned = sum of expanded degrees.
 * `queries`: nq rows at `qstride_floats` stride holding at least `dim` floats each; for COSINE the
 * caller normalises queries first, as the harness does (tests/test_search_roargraph.cpp:167-172).
 * Host-buffer form: the kernel reads the query rows from pinned host memory in place (the caller's block if it is
 * pinned, see roar_host_alloc; else an internal pinned copy), results return with one asynchronous copy.  Columns
 * [dim, padded dim) of a query row are never read from the caller (they are zero filled).  Any of
 * cmps/hops/scanned may be NULL.
 * Safe to call from several host threads on one index (calls are serialised), like the reference's
 * re-entrant SearchRoarGraph; the device-buffer form below is one host thread per index.
 * Returns ROAR_ERR_SHORT if some query ended with fewer than k results (their tail ids are
 * 0xFFFFFFFF); the other queries' results are still valid. */
int roar_search_batch(roar_index *idx, const float *queries, uint64_t nq, uint32_t qstride_floats, uint32_t k,
                      uint32_t L_pq, uint32_t *ids, float *dists, uint32_t *cmps, uint32_t *hops,
                      uint32_t *scanned);
/* Device-buffer form: every pointer is HBM-resident, launch is asynchronous on `hip_stream`
 * (a hipStream_t, NULL = default stream); no implicit synchronisation.  Short results are reported
 * through *short_count_dev (u32, may be NULL).
 * Launches of one index on DIFFERENT streams may overlap (each takes a 16-byte launch cell from a ring of 256, zero on entry and
 * zeroed again by its last workgroup, which also leaves the launch's stamp in it): at most 256 launches of one index may be in flight
 * at a time.  A launch that finds its cell NOT as its predecessor should have left it -- that launch faulted, or the ring went round
 * under it -- does not search: it stores 0xFFFFFFFF to *short_count_dev and ends (pass short_count_dev to see it); the host-buffer
 * calls return ROAR_ERR_DEVICE for it and put the cells back themselves, a caller of this form does so with option
 * "reset_launch_cells" once nothing of the index is in flight.  Launches that use per-workgroup lists in HBM (visited_mode
 * EXACT, or LDS with spill lists) are ordered one behind the other by an event.  "last_kernel_ns" is the duration of the most
 * recently ENQUEUED launch between its own two events on its own stream; with launches overlapping it includes what that stream
 * waited for -- use one launch at a time (or events of your own) to time a kernel.  The first search of a kind calibrates the launch
 * plan (option "calibrate") and synchronises `hip_stream` once. */
int roar_search_batch_device(roar_index *idx, const float *queries_dev, uint64_t nq, uint32_t qstride_floats,
                             uint32_t k, uint32_t L_pq, uint32_t *ids_dev, float *dists_dev, uint32_t *cmps_dev,
                             uint32_t *hops_dev, uint32_t *scanned_dev, uint32_t *short_count_dev,
                             void *hip_stream);

/* ---- a stream of batches (device buffers) -----------------------------------------------------------------------------------------
 * Replaces the harness loop tests/test_search_roargraph.cpp:202-212 when batches follow one another -- a serving loop, or one GPU's
 * share of every batch at N = 8 (1 250 of 10 000 queries, BASELINE configs[3]): a launch per batch ends with its slowest query
 * while most of the chip idles.  Here a batch is not a launch.  submit() PUBLISHES the batch (a 64-byte record in pinned host memory)
 * and returns a ticket; workgroups of the batch-stream form of the search kernel take queries from the published batches in batch
 * order, cross batch boundaries, and leave only when nothing published is left (an open but idle stream holds no part of the chip,
 * and nothing on the GPU ever waits for the host); submit() starts workgroups when fewer are alive than the chip holds and queries are
 * waiting.  wait() returns when every query of the ticket's batch has its results in device memory -- readable from then on by a
 * hipMemcpy or by any kernel launched afterwards.  Results per batch are bit-identical to roar_search_batch_device (same search code;
 * cmps as with visited_mode LDS_FORGETFUL: >= the reference's).
 *   open    k, L_pq for every batch of the stream.  visited_mode must be AUTO or LDS_FORGETFUL (workgroups of several launches live
 *           side by side: nothing may be indexed by workgroup in HBM); the launch plan (screen, table size) is what the index's
 *           calibration measured for the largest batches at this L_pq -- measured on the first batch if it has not been yet, PROVIDED
 *           that batch is a sample one can measure on (at least 12 queries per compute unit: less than one residency of the chip
 *           makes every table tie and lets noise pick).  A stream whose first batches are smaller runs on the default plan and
 *           caches nothing; the first batch that is large enough (or a calibration made meanwhile by other launches of the index)
 *           drains the stream once and plans it again (info keys "plan_provisional", "replans").  Results never depend on the plan.
 *   submit  as roar_search_batch_device without a stream argument: the batch must be COMPLETE in device memory when submit is called
 *           (synchronise whatever wrote it) and must stay valid and unchanged until its ticket has been waited for; the result
 *           arrays likewise.  At most 256 batches may be incomplete at a time (submit blocks on the oldest beyond that).  nq = 0:
 *           ticket 0, nothing to wait for.  Thread-safe; several host threads may submit to and wait on one stream.
 *   wait    blocks (polling pinned memory) until the batch is complete; *short_count (may be NULL) = queries of the batch that ended
 *           with fewer than k results (tail ids 0xFFFFFFFF).  Tickets may be waited for in any order, more than once, or never.
 *           (The stream's memory does not grow with the batches it has served: full per-ticket records are kept for the running epoch
 *           of 65 536 batches and the one before it; of an older ticket only a non-zero short count is kept, its times() are gone.)
 *           A batch that does not complete within 60 s fails the stream (ROAR_ERR_DEVICE; every later call fails).
 *   times   device wall clock (ns since an arbitrary origin) at which the batch's first query was taken and its last query finished.
 *   close   waits for every submitted batch and for the stream's launches to end, then frees the stream.  A FAILED stream: its launches
 *           are given 10 s to end (a workgroup that waits for another gives up after 2 s on the device); if they did, the stream is
 *           freed and the index released, and close() returns the error; if they did not, NOTHING is freed and the index stays
 *           pinned -- roar_destroy / roar_set_base / a new graph keep refusing it, because they would free what those workgroups read.
 * While a stream is open the index's base rows and graph must not be replaced (those calls fail with ROAR_ERR_ARG), and other
 * searches on the same device share the chip with the stream's workgroups.
 * info keys: "launches" (kernel launches so far), "workgroups_launched", "workgroups_alive", "resident_capacity", "heals" (times
 * wait() found a batch pending and no workgroup alive), "epochs", "replans", "plan_provisional", "batches_submitted", "batches_completed", "ring", "lds_bytes",
 * "screen", "vis_entries", "tag_protect", "register_pool", "wall_clock_khz"; "take_calls" / "take_ns_total" / "busy_ns_total": calls of
 * the kernel's dequeue, device time all workgroups together spent in it / in the searches between (this epoch; one device read each);
 * "debug_break" (tests): marks the stream failed, as a failed launch or a timed-out wait would. */
typedef struct roar_stream roar_stream;
int roar_search_stream_open(roar_index *idx, uint32_t k, uint32_t L_pq, roar_stream **out);
int roar_search_stream_submit(roar_stream *s, const float *queries_dev, uint64_t nq, uint32_t qstride_floats, uint32_t *ids_dev,
                              float *dists_dev, uint32_t *cmps_dev, uint32_t *hops_dev, uint32_t *scanned_dev, uint64_t *ticket);
int roar_search_stream_wait(roar_stream *s, uint64_t ticket, uint32_t *short_count);
int roar_search_stream_times(roar_stream *s, uint64_t ticket, uint64_t *first_ns, uint64_t *done_ns);
int roar_search_stream_info(roar_stream *s, const char *key, int64_t *value);
int roar_search_stream_close(roar_stream *s);

/* ---- replicated index, one query block split over g devices (BASELINE configs[3]) ---------
 * The reference has ONE query set and one timed loop over it (tests/test_search_roargraph.cpp:202-209); queries share
 * only read-only state (src/index_bipartite.cpp:2311-2420 has no cross-query dependence), so the block shards by
 * query with no collective.  roar_clone_to_device makes a replica of an index (base rows + projection graph + options)
 * on another device (device-to-device copy; the same device is allowed).  roar_search_batch_multi gives replica r the
 * queries [r * nq / g, (r + 1) * nq / g) -- one host thread + one stream per replica -- and every replica writes its
 * own rows of the caller's arrays; arguments and errors otherwise as roar_search_batch (ROAR_ERR_SHORT if any slice
 * had a short query, the message of the first one).  Results do not depend on g. */
int roar_clone_to_device(roar_index *src, int device, roar_index **out);
int roar_search_batch_multi(roar_index *const *replicas, uint32_t g, const float *queries, uint64_t nq,
                            uint32_t qstride_floats, uint32_t k, uint32_t L_pq, uint32_t *ids, float *dists, uint32_t *cmps,
                            uint32_t *hops, uint32_t *scanned);

/* Pinned, device-mapped host memory for query blocks (what load_data / data_align of the facade header return):
 * roar_search_batch reads a query block that lives in such memory in place -- no staging copy, no host-to-device
 * copy in front of the kernel -- when its rows are strided at >= the padded dimension and the dimension needs no
 * padding; any other host pointer works too: its rows are copied into an internal pinned block by the calling thread
 * WHILE the kernel already runs (the kernel takes queries in row order and waits for the rows-written counter), so the
 * block must stay valid and unchanged until the call returns, as for any argument.  NULL on failure. */
void *roar_host_alloc(uint64_t bytes);
void roar_host_free(void *p);
/* The same for a block the caller allocated itself (malloc / memalign / new[], e.g. what the reference's load_data /
 * data_align return, util.h:64,196): page-lock and map it (hipHostRegister) so that roar_search_batch reads it in place;
 * unregister before releasing the block. */
int roar_host_register(void *p, uint64_t bytes);
int roar_host_unregister(void *p);

/* ---- bipartite two-hop variant (the paper's ablation baseline) -------------------
 * Graph over base nodes [0, n) and training-query nodes [n, n_total); rows of base nodes hold
 * training-query nodes and the reverse.  On-disk format of IndexBipartite::Save / Load
 * (src/index_bipartite.cpp:2045-2071): u32 npts; npts x {u32 deg; u32 nbr[deg]} (no entry-point word). */
int roar_load_bipartite_file(roar_index *idx, const char *path); /* IndexBipartite::Load  cpp:2060-2071 */
int roar_save_bipartite_file(roar_index *idx, const char *path); /* IndexBipartite::Save  cpp:2045-2058 */
int roar_set_bipartite_csr(roar_index *idx, uint64_t n_total, const uint64_t *offsets, const uint32_t *nbrs);
/* two-phase like roar_get_graph_csr: offsets/nbrs may be NULL to size the buffers first */
int roar_get_bipartite_csr(roar_index *idx, uint64_t *n_total, uint64_t *nedges, uint64_t *offsets, uint32_t *nbrs);

/* BuildBipartite -> qbaseNNbipartite  src/index_bipartite.cpp:42-141, 235-280 (driver
 * tests/test_only_build_bipartite.cpp:100-135): from the learn->base kNN lists (host, [n_learn][K], nearest
 * first) keep min(K, M_pjbp) ids per training query; the nearest is its target and gains the edge
 * target -> n + sq, the others form the row of node n + sq.  Target rows are filled in ascending sq
 * (the reference's order depends on its OpenMP lock order; the search result does not depend on it). */
int roar_build_bipartite(roar_index *idx, const uint32_t *learn_knn, uint64_t n_learn, uint32_t K, uint32_t M_pjbp);

/* Entry points.  The reference draws 10 uniform base ids per query from std::random_device
 * (src/index_bipartite.cpp:287-294), so two runs of it differ; here the ids are an argument
 * ([nq][n_init], 1 <= n_init <= 64) and this helper is the counter-based generator the host API uses
 * when the caller passes NULL (seed = option "bipartite_seed", advanced after every such call). */
int roar_bipartite_entry_ids(uint64_t seed, uint64_t nq, uint32_t n_init, uint64_t n_base, uint32_t *out);

/* IndexBipartite::SearchBipartiteGraph  src/index_bipartite.cpp:282-356 under the harness loop
 * tests/test_search_bipartite.cpp:194-201, batched.  Every entry point is scored, inserted and marked
 * visited (not counted in cmps); an expansion scores the not-yet-visited base nodes two hops away
 * (first occurrence wins); hops counts expansions, including those of nodes with an empty row.
 * Errors as roar_search_batch ("not enough results: x, expected: k" -> ROAR_ERR_SHORT).
 * With "visited_mode" AUTO a query is run by eight waves up to L_pq = 200 (work per query is very uneven on this
 * graph), by one wave beyond; an explicit visited_mode keeps the one-wave form.  Results do not depend on it. */
int roar_search_bipartite_batch(roar_index *idx, const float *queries, uint64_t nq, uint32_t qstride_floats,
                                uint32_t k, uint32_t L_pq, const uint32_t *init_ids, uint32_t n_init, uint32_t *ids,
                                float *dists, uint32_t *cmps, uint32_t *hops);
int roar_search_bipartite_batch_device(roar_index *idx, const float *queries_dev, uint64_t nq,
                                       uint32_t qstride_floats, uint32_t k, uint32_t L_pq,
                                       const uint32_t *init_ids_dev, uint32_t n_init, uint32_t *ids_dev,
                                       float *dists_dev, uint32_t *cmps_dev, uint32_t *hops_dev,
                                       uint32_t *scanned_dev, uint32_t *short_count_dev, void *hip_stream);

/* ---- distance kernels alone -----------------------------------------------------------------
 * DistanceL2::compare / DistanceInnerProduct::compare  include/efanna2e/distance.h:39-89,179-223.
 * out[q * m + j] = distance(base[ids[q * m + j]], queries[q]) with the index's metric, computed
 * by the same device routine the search uses (bit-identical to the reference's AVX-512 + FMA
 * summation order).  Host buffers. */
int roar_distances(roar_index *idx, const float *queries, uint64_t nq, uint32_t qstride_floats,
                   const uint32_t *ids, uint32_t m, float *out);

/* ---- exact kNN generator ---------------------------------------------------------
 * Producer of the two exact-kNN inputs the reference consumes but cannot make itself (their tool
 * lived in the empty thirdparty/DiskANN submodule): the query->base ground truth read by
 * load_gt_meta / load_gt_data_with_dist (include/efanna2e/util.h:84-155) and the learn->base
 * top-M_sq lists read by LoadLearnBaseKNN (src/index_bipartite.cpp:2622-2639).  Brute force over the
 * loaded base rows with the index metric: f32 MFMA GEMM + threshold top-k, ids sorted by
 * (distance, id); distances are reference-signed (IP: negated dot; L2: squared, via |q|^2 - 2qx + |x|^2,
 * so they agree with DistanceL2::compare to rounding, not bit for bit).  Device buffers; k <= 128;
 * dists_dev may be NULL.  Synchronises the stream before returning. */
int roar_exact_knn_device(roar_index *idx, const float *queries_dev, uint64_t nq, uint32_t qstride_floats, uint32_t k,
                          uint32_t *ids_dev, float *dists_dev, void *hip_stream);

/* Host-buffer form of the same (queries: nq rows of `dim` valid floats; ids [nq][k], dists [nq][k] or NULL). */
int roar_exact_knn(roar_index *idx, const float *queries, uint64_t nq, uint32_t qstride_floats, uint32_t k, uint32_t *ids,
                   float *dists);

/* ---- top-k merge for the index-sharded configuration ------------------------------------------
 * Not in the reference (new capability): merge `nshards` per-shard sorted top-k lists
 * per query by (distance, id) into a global top-k.  Device buffers laid out [shard][nq][k]. */
int roar_merge_topk_device(int device, uint32_t nshards, uint64_t nq, uint32_t k, const uint32_t *ids_dev,
                           const float *dists_dev, uint32_t *out_ids_dev, float *out_dists_dev, void *hip_stream);

/* local ids of one shard -> global ids: ids_dev[i] += base for every entry that is a result (0xFFFFFFFF = "no result" is
 * kept).  Device buffer, asynchronous on `hip_stream`. */
int roar_rebase_ids_device(int device, uint32_t *ids_dev, uint64_t count, uint32_t base, void *hip_stream);

/* ---- index-sharded search (BASELINE configs[4]: the base set exceeds one GPU) -------------
 * Not in the reference, whose one index lives in one address space; this is what its harness loop
 * (tests/test_search_roargraph.cpp:202-209) calls when the base rows are cut into contiguous id ranges, one RoarGraph per
 * range (local ids, own entry point).  One process drives all devices:
 *   roar_sharded_create     n_shards (<= 32) empty shard indexes, shard s on HIP device devices[s] (a device may hold
 *                           several shards; ranks = distinct devices, and ranks x shards-per-rank <= 32)
 *   roar_sharded_shard      the roar_index of shard s -- load its base rows and its graph with the ordinary calls
 *                           (roar_load_base_fbin / roar_set_base*, roar_load_graph_file / roar_build_roargraph ...); owned
 *                           by the roar_sharded
 *   roar_sharded_set_id_base  global id of shard s's row 0 (default: the running sum of the shard sizes)
 *   roar_search_batch_sharded  every query on every shard with the same L_pq (one host thread + one stream per device,
 *                           shards of a device back to back), local ids rebased to global ids, then the ONE exchange
 *                           step of the path -- ncclAllGather (RCCL over xGMI; ncclCommInitAll on first use) of every
 *                           rank's [shard][nq][k] ids and distances, nq x k x 8 bytes per shard -- and the k-way merge by
 *                           (distance, id) on every rank (roar_merge_topk_device); rank 0's lists return in the caller's
 *                           HOST arrays ids[nq][k] (global ids), dists[nq][k]; cmps / hops (may be NULL) are the query's
 *                           distance evaluations / expansions summed over the shards.  Errors as roar_search_batch;
 *                           ROAR_ERR_SHORT only if all shards together returned fewer than k results for some query.
 *   roar_sharded_info       "shards", "ranks", "last_search_ns" (slowest rank: query upload + its shards' searches),
 *                           "last_exchange_ns" (slowest rank: all-gather + merge), both by HIP events of the last call */
typedef struct roar_sharded roar_sharded;
int roar_sharded_create(roar_sharded **out, uint32_t dim, int metric, const int *devices, uint32_t n_shards);
void roar_sharded_destroy(roar_sharded *sh);
roar_index *roar_sharded_shard(roar_sharded *sh, uint32_t s);
int roar_sharded_set_id_base(roar_sharded *sh, uint32_t s, uint64_t first_global_id);
int roar_search_batch_sharded(roar_sharded *sh, const float *queries, uint64_t nq, uint32_t qstride_floats, uint32_t k,
                              uint32_t L_pq, uint32_t *ids, float *dists, uint32_t *cmps, uint32_t *hops);
int roar_sharded_info(roar_sharded *sh, const char *key, int64_t *value);

/* ---- options / introspection ----------------------------------------------------------------
 * Parameters::Set / Get  include/efanna2e/parameters.h:15-57 carry L_pq per call in the reference;
 * here L_pq is an argument and the remaining knobs are integers:
 *   "visited_mode"   ROAR_VISITED_EXACT | ROAR_VISITED_LDS | ROAR_VISITED_LDS_FORGETFUL | ROAR_VISITED_AUTO (default AUTO)
 *   "waves_per_cu"   resident single-wave workgroups per CU to launch (0 = auto: what registers and LDS allow)
 *   "vis_entries"    LDS visited-table entries per query (0 = auto from L_pq and the graph; 2^k, 3 * 2^k or 5 * 2^k)
 *   "max_grid"       cap on the number of resident workgroups launched (0 = no cap)
 *   "edge_floats"    0 auto | 1 off: inner product over rows whose screening codes fill whole 128-byte lines (128-d, 512-d): the scale
 *                    and bounds of every neighbour's screening row are kept beside the adjacency (8 bytes per edge, made on first
 *                    use) and travel with the adjacency row, so that a screened neighbour costs its code row's lines and nothing
 *                    else; results do not depend on it (info key "last_edge_floats")
 *   "reset_launch_cells"  (any value) puts the launch cells back after a launch of the device-buffer form reported 0xFFFFFFFF in
 *                    *short_count_dev; nothing of the index may be in flight.  ("debug_dirty_launch_cell": tests -- leaves the next
 *                    launch's cell as a launch that died would have.)
 *   "search_kernel"  0 auto | 1 one wave per query (throughput form) | 2 four waves per query (latency form) |
 *                    3 two waves per query; auto picks four waves for batches of at most one query per CU on a dense
 *                    graph (average degree x row bytes >= 24 x 800), one wave otherwise
 *   "pool_form"      0 auto (candidate pool in a register pair of wave 0 for L_pq <= 64, in LDS beyond) | 1 always LDS
 *   "hub_mode"       0 auto | 1 off | 2 on: the 65 535 highest-in-degree nodes keep their visited bits in an exact
 *                    on-chip bitmap (a hub-annotated copy of the adjacency is made on first use; graphs with a row
 *                    longer than 170 neighbours do without); auto = for L_pq >= 100 on dense graphs (same measure);
 *                    results do not depend on it
 *   "screen"         0 auto | 1 off | 2 on: the index keeps a second copy of the base rows in 8-bit codes (made when the base rows are set:
 *                    200-d rows 256 bytes = two 128-byte lines instead of seven, 512-d rows 512 + 16 bytes = four lines and a 16-byte read instead of
 *                    sixteen) and a hop scores its unvisited neighbours on it first; a neighbour whose exact key is CERTAINLY
 *                    above the pool's reject threshold (rigorous bound: this row's quantisation error norm + the rounding of
 *                    the screen's own arithmetic) is dropped there, the others are scored on the f32 rows as before -- ids,
 *                    distances, hops and cmps do not depend on it (csrc/roar_search_v2.h, screen8_pass).  Kernels: inner
 *                    product / cosine at 128, 200 and 512 floats per row, L2 at 128 and 512; auto = decided by measurement on the loaded
 *                    index and the caller's own queries (option "calibrate": the bound's power to drop rows is a property of the data --
 *                    on sets whose scores are bunched within the bound's slack it passes nearly everything and costs a second
 *                    dependent round trip for nothing), 2 = on wherever there is a kernel
 *   "calibrate"      1 (default): what no option fixes -- screen on / off, visited form, table size -- is MEASURED on the loaded index:
 *                    the first search of a (L_pq band, batch-size band, grid cap) times 5-10 candidate plans (tag tables from 16 384
 *                    down to 1 024 tags and the 3 x 2^k sizes between them, HBM tags for long pools) on the first (at most
 *                    12 288) queries of that call, results into scratch buffers, and keeps the fastest for the launches that follow
 *                    (that call synchronises its stream; results never depend on the plan; ROAR_LOG_CALIBRATION=1 in the environment
 *                    prints every decision).  L_pq bands: half-octaves up to 512, 160 entries (one 1 280-byte LDS block of pool) beyond;
 *                    launches that share the chip (grid cap, stream) have entries of their own, ranked by resident workgroups among
 *                    candidates within 2 % of the fastest; a cached table steps down when a longer pool of its band would otherwise
 *                    keep fewer workgroups resident than it was measured with; 0: fixed defaults (screen on for rows of 768 bytes and more, the on-chip table with
 *                    4 096 / 8 192 tags up to L_pq = 1 536, HBM tags beyond); 2: forget what was measured so far
 *   "tag_protect"    forgetful on-chip tag table under a candidate pool in LDS (L_pq > 64, one wave per query, device or pinned queries):
 *                    1 = a tag whose node is in the pool is never the victim of an insertion (bit 15 of the tag, kept by the queue's
 *                    merge; needs base points / sets < 32 767) -- a forgotten pool member is the expensive revisit: it always passes
 *                    the 8-bit screen and costs an f32 row and a merge candidate; 0 = plain replacement; -1 (default) = measured by
 *                    the calibration, on without one.  Results do not depend on it; cmps moves towards the reference's
 *   "knn_mode"       roar_exact_knn(_device): 0 (default) bf16 MFMA screen with a rigorous per-row bound + exact f32 re-rank in the
 *                    reference's summation order (csrc/roar_knn_bf16.hip; it hands over to form 1 when the bound cannot be claimed),
 *                    1 every score by f32 MFMA (csrc/roar_knn.hip)
 *   "profile_phases" 1: instrumented kernel (200-d inner-product and 128-d L2 shapes), see roar_last_profile
 *   "bipartite_seed" seed of the entry-point generator of roar_search_bipartite_batch(init_ids = NULL)
 *   "combine_single_queries"  1 (default): roar_search_batch calls with nq = 1 that arrive from several host
 *                    threads while a launch is in flight share the next launch (the reference's harness calls
 *                    SearchRoarGraph once per query from num_threads OpenMP threads,
 *                    tests/test_search_roargraph.cpp:203-209); 0: one launch per call
 * roar_get_info keys: "n", "dim", "dim_padded", "ep", "max_degree", "nedges", "adj_stride",
 *   "bipartite_nodes", "bipartite_edges", "last_combined" (queries carried by the most recent combined launch),
 *   "row_stride", "graph_hash" (64-bit hash of the device rows and the entry point), "last_grid", "last_workgroups_per_cu" (what the last plan keeps resident per CU, before the grid is cut to the batch or a cap), "last_lds_bytes",
 *   "last_vis_entries", "last_waves_per_query", "last_register_pool", "last_hub_bitmap", "last_screen", "screen_row_bytes" (bytes per
 *   row of the 8-bit screening copy at this dimension), "last_visited_forgetful", "last_tag_protect", "last_plan_calibrated" (1 if the most recent search ran a measured plan),
 *   "calibrations" (launch kinds measured so far), "last_knn_bf16" (1 if the most recent exact kNN call ran the bf16-screened form), "last_knn_rerank_rows" (rows it scored on f32), "last_kernel_ns"
 *   (duration of the most recent search kernel, measured with HIP events on its own stream;
 *   reading it synchronises on those events). */
int roar_set_option(roar_index *idx, const char *key, int64_t value);
int roar_get_info(roar_index *idx, const char *key, int64_t *value);

/* Diagnostics (no reference counterpart): with option "profile_phases" = 1 the search runs an instrumented
 * instantiation of the same kernel; this copies out, for the first nq queries of the most recent such launch,
 * u64[nq][16] = shader-clock cycles of {whole query, select, adjacency fetch, visited filter, row scoring, merge},
 * speculative adjacency fetches that were used, adjacency round trips paid, then wave 0's scoring passes: cycles
 * {issuing loads, waiting + FMAs, reduction + packing}, number of passes, candidates that reached the merge; 3 spare. */
int roar_last_profile(roar_index *idx, uint64_t *out, uint64_t nq);

/* Measurement (no reference counterpart; SURVEY 8(d) prices a query by the counters the reference returns, cmps and hops,
 * src/index_bipartite.cpp:2397,2419 -- with the 8-bit screen most of those distance evaluations never read their f32 row, so the
 * kernel also counts what it really asks the memory side for): while a block is set, every SearchRoarGraph launch of this index
 * with nq <= capacity_queries writes, for query q, counters_dev[4 q ..] = {rows scored on the 8-bit screening copy, rows scored
 * on the f32 rows (entry point included; equals cmps + 1 with the screen off), adjacency words fetched (whole 64-word chunks,
 * speculative next-row fetches that were dropped included), 0}.  Device memory, 16-byte aligned, caller owned; NULL switches
 * it off.  Launches that overlap write the same rows: set it for untimed, one-at-a-time launches.  The counting form exists for
 * one wave per query with device-resident or pinned queries: any other launch fails with ROAR_ERR_ARG while a block is set
 * (option "search_kernel" = 1).  Hub-bitmap form: a row fetch is four chunks (ids + bitmap slots). */
int roar_set_traffic_counters_device(roar_index *idx, uint32_t *counters_dev, uint64_t capacity_queries);

/* Testing hook for the error paths: the n-th device allocation the library makes from now on fails as if HBM were
 * exhausted (n <= 0 disarms; ROAR_FAIL_AT=n in the environment arms it at load).  Returns how many allocations were still
 * to go on the previous arming (0: it was disarmed or has fired).  tests/test_gpu_parity.py walks n over load / search /
 * clone scenarios and checks that every failure is either reported as ROAR_ERR_DEVICE or absorbed by a documented
 * fallback, and leaves no device memory behind. */
int roar_debug_fail_device_alloc_at(int64_t n);

/* Known-answer surface for the 8-bit screen (csrc/roar_search_v2.h, screen8i_pass / screen8_pass; no reference counterpart -- the
 * screen is this library's own device): runs the search kernels' screening passes over explicit pairs, row ids[q][c] of query q
 * against its OWN threshold thresholds[q][c], and returns keeps[q][c] = 1 if the screen would pass the row on to the exact scoring,
 * 0 if it would drop it.  The property the search relies on: with thresholds[q][c] = the row's exact distance (roar_distances),
 * every verdict is 1 -- a row is only ever dropped when its exact distance is certainly ABOVE the threshold.  Host arrays,
 * [nq][m]; shapes with a screening kernel only (option "screen"); makes the screening copy if the index has none yet. */
int roar_debug_screen_keeps(roar_index *idx, const float *queries, uint64_t nq, uint32_t qstride_floats, const uint32_t *ids,
                            const float *thresholds, uint32_t m, uint8_t *keeps);

/* Known-answer surface for NeighborPriorityQueue (include/efanna2e/neighbor.h:138-223): replays a scripted sequence on
 * the device with the queue routines of the search kernels -- ops[i] == 0: insert(ids[i], dists[i]) (:150-183), else
 * closest_unexpanded() if has_unexpanded_node() (:185-194) -- and returns the state after every op,
 * snap[i] = {size, cur, (id, distance bits, flag) x capacity} (u32[n_ops][2 + 3 capacity], unused cells 0), the layout
 * of tests/golden/pq_trace.npz.  pool_in_registers: 0 = the LDS pool (any L_pq), 1 = the register pool (L_pq <= 64).
 * Host arrays; capacity <= 64. */
int roar_queue_trace(int device, uint32_t capacity, uint32_t n_ops, const uint8_t *ops, const uint32_t *ids, const float *dists,
                     int pool_in_registers, uint32_t *snap);

/* InitVisitedListPool(num_threads)  include/index_bipartite.h:133: sizes the EXACT-mode tag pool
 * (one u8[n] list per resident wave).  Optional: done lazily by the first EXACT search. */
int roar_init_visited_pool(roar_index *idx, uint32_t slots);

#ifdef __cplusplus
}
#endif
#endif /* ROARGRAPH_AMD_H */
