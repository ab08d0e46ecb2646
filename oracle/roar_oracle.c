/*
 * roar_oracle.c -- CPU restatement of the RoarGraph projected-graph beam search.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under mysteryann_amd/ may include, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / reported CPU baseline.
 *
 * Plain C restatement (no reference source is included or copied) of:
 *   A1  IndexBipartite::SearchRoarGraph      /root/reference/src/index_bipartite.cpp:2311-2420
 *   A2  DistanceInnerProduct::compare        /root/reference/include/efanna2e/distance.h:179-223
 *   A3  DistanceL2::compare                  /root/reference/include/efanna2e/distance.h:39-89
 *   A4  Neighbor / NeighborPriorityQueue     /root/reference/include/efanna2e/neighbor.h:21-34,138-223
 *   A5  VisitedList (u16 epoch tags)         /root/reference/include/visited_list_pool.h:8-29
 *   A7  graph / fbin / gt file formats       index_bipartite.cpp:2097-2117,2606-2619; util.h:84-211
 *   A8  row padding + normalize              util.h:37-75,179-225
 *
 * Parity pinning: the pieces the reference lets us compile without stand-in
 * headers (A2-A5, A7/A8 loaders: all header-only, stub-free) are compiled as
 * they lie into oracle/_ref/ (see oracle/ref_driver.cpp, oracle/Makefile) and
 * this file is checked bit-for-bit against them in tests/test_oracle_pinning.py
 * and through the committed fixtures in tests/golden/.  The A1 loop itself
 * lives in index_bipartite.cpp, which needs Boost + robin-map headers that are
 * absent from this image, so the loop is restated in both places.
 *
 * Arithmetic note (A2/A3): the reference is built with -Ofast -march=native
 * (CMakeLists.txt:28); gcc contracts every mul+add of the AVX-512 code into
 * vfmadd231ps, so each of the 16 SIMD lanes is an fmaf chain over i = lane,
 * lane+16, ...; the tails and the reduction tree follow distance.h line by line.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_METRIC_L2 0     /* distance.h:15 */
#define ORACLE_METRIC_IP 1
#define ORACLE_METRIC_COSINE 4 /* scored with IP after normalisation, src/index.cpp:16-19 */

/* ---------------------------------------------------------------- A3 */
float oracle_dist_l2(const float *x, const float *y, unsigned d) {
    float s0[16], s1[8], s2[4];
    for (int j = 0; j < 16; ++j) s0[j] = 0.0f;
    while (d >= 16) { /* distance.h:42-50 */
        for (int j = 0; j < 16; ++j) {
            float t = x[j] - y[j];
            s0[j] = fmaf(t, t, s0[j]);
        }
        x += 16; y += 16; d -= 16;
    }
    for (int j = 0; j < 8; ++j) s1[j] = s0[j + 8] + s0[j]; /* :52-53 */
    if (d >= 8) {                                          /* :55-63 */
        for (int j = 0; j < 8; ++j) {
            float t = x[j] - y[j];
            s1[j] = fmaf(t, t, s1[j]);
        }
        x += 8; y += 8; d -= 8;
    }
    for (int j = 0; j < 4; ++j) s2[j] = s1[j + 4] + s1[j]; /* :65-66 */
    if (d >= 4) {                                          /* :68-76 */
        for (int j = 0; j < 4; ++j) {
            float t = x[j] - y[j];
            s2[j] = fmaf(t, t, s2[j]);
        }
        x += 4; y += 4; d -= 4;
    }
    if (d > 0) { /* masked_read, :78-83: lanes >= d read as 0 */
        for (unsigned j = 0; j < 4; ++j) {
            float t = (j < d ? x[j] : 0.0f) - (j < d ? y[j] : 0.0f);
            s2[j] = fmaf(t, t, s2[j]);
        }
    }
    /* two hadd_ps, :85-86 */
    float h0 = s2[0] + s2[1], h1 = s2[2] + s2[3];
    return h0 + h1;
}

/* ---------------------------------------------------------------- A2 */
float oracle_dist_ip(const float *a, const float *b, unsigned d) {
    float s0[16], s1[8], s2[4];
    for (int j = 0; j < 16; ++j) s0[j] = 0.0f;
    while (d >= 16) { /* distance.h:182-189 */
        for (int j = 0; j < 16; ++j) s0[j] = fmaf(a[j], b[j], s0[j]);
        a += 16; b += 16; d -= 16;
    }
    for (int j = 0; j < 8; ++j) s1[j] = s0[j + 8] + s0[j]; /* :191-192 */
    if (d >= 8) {                                          /* :194-201 */
        for (int j = 0; j < 8; ++j) s1[j] = fmaf(a[j], b[j], s1[j]);
        a += 8; b += 8; d -= 8;
    }
    for (int j = 0; j < 4; ++j) s2[j] = s1[j + 4] + s1[j]; /* :203-204 */
    if (d >= 4) {                                          /* :206-213 */
        for (int j = 0; j < 4; ++j) s2[j] = fmaf(a[j], b[j], s2[j]);
        a += 4; b += 4; d -= 4;
    }
    if (d > 0) { /* :215-219 */
        for (unsigned j = 0; j < 4; ++j)
            s2[j] = fmaf(j < d ? a[j] : 0.0f, j < d ? b[j] : 0.0f, s2[j]);
    }
    float h0 = s2[0] + s2[1], h1 = s2[2] + s2[3];
    return -(h0 + h1); /* :223, "-1.0 *" */
}

float oracle_dist(int metric, const float *a, const float *b, unsigned d) {
    return metric == ORACLE_METRIC_L2 ? oracle_dist_l2(a, b, d) : oracle_dist_ip(a, b, d);
}

/* ---------------------------------------------------------------- A4 */
typedef struct {
    uint32_t id;
    float distance;
    uint8_t flag;
} oracle_nbr; /* neighbor.h:21-24 */

static inline int nbr_less(const oracle_nbr *a, const oracle_nbr *b) { /* neighbor.h:29-31 */
    return a->distance < b->distance || (a->distance == b->distance && a->id < b->id);
}

typedef struct {
    size_t size, capacity, cur;
    oracle_nbr *data; /* capacity + 1 slots, neighbor.h:142 */
} oracle_pq;

void oracle_pq_init(oracle_pq *q, size_t capacity) {
    q->size = 0; q->capacity = capacity; q->cur = 0;
    q->data = (oracle_nbr *)calloc(capacity + 1, sizeof(oracle_nbr));
}
void oracle_pq_free(oracle_pq *q) { free(q->data); q->data = NULL; }

void oracle_pq_insert(oracle_pq *q, uint32_t id, float distance) { /* neighbor.h:150-183 */
    oracle_nbr nbr = {id, distance, 0};
    if (q->size == q->capacity && nbr_less(&q->data[q->size - 1], &nbr)) return;
    size_t lo = 0, hi = q->size;
    while (lo < hi) {
        size_t mid = (lo + hi) >> 1;
        if (nbr_less(&nbr, &q->data[mid])) hi = mid;
        else if (q->data[mid].id == nbr.id) return;
        else lo = mid + 1;
    }
    if (lo < q->capacity) memmove(&q->data[lo + 1], &q->data[lo], (q->size - lo) * sizeof(oracle_nbr));
    q->data[lo] = nbr;
    if (q->size < q->capacity) q->size++;
    if (lo < q->cur) q->cur = lo;
}

oracle_nbr oracle_pq_closest_unexpanded(oracle_pq *q) { /* neighbor.h:185-192 */
    q->data[q->cur].flag = 1;
    size_t pre = q->cur;
    while (q->cur < q->size && q->data[q->cur].flag) q->cur++;
    return q->data[pre];
}

/* Scripted trace driver used to compare with the reference queue op by op.
 * ops[i] = 0: insert (ids[i], dists[i]); 1: closest_unexpanded.
 * After every op writes a snapshot row: [size, cur, then capacity x (id, dist bits, flag)]. */
void oracle_pq_trace(uint32_t capacity, uint32_t n_ops, const uint8_t *ops, const uint32_t *ids,
                     const float *dists, uint32_t *snap /* n_ops x (2 + 3*capacity) */) {
    oracle_pq q;
    oracle_pq_init(&q, capacity);
    size_t row = 2 + 3 * (size_t)capacity;
    for (uint32_t i = 0; i < n_ops; ++i) {
        if (ops[i] == 0) oracle_pq_insert(&q, ids[i], dists[i]);
        else if (q.cur < q.size) (void)oracle_pq_closest_unexpanded(&q);
        uint32_t *s = snap + (size_t)i * row;
        memset(s, 0, row * sizeof(uint32_t));
        s[0] = (uint32_t)q.size; s[1] = (uint32_t)q.cur;
        for (size_t j = 0; j < q.size; ++j) {
            s[2 + 3 * j] = q.data[j].id;
            memcpy(&s[3 + 3 * j], &q.data[j].distance, 4);
            s[4 + 3 * j] = q.data[j].flag;
        }
    }
    oracle_pq_free(&q);
}

/* ---------------------------------------------------------------- A5 */
typedef struct {
    uint16_t curV;
    uint16_t *mass;
    uint64_t n;
} oracle_visited; /* visited_list_pool.h:8-29 */

static void visited_init(oracle_visited *v, uint64_t n) {
    v->curV = (uint16_t)-1; v->n = n;
    v->mass = (uint16_t *)malloc(sizeof(uint16_t) * (n ? n : 1));
}
static void visited_reset(oracle_visited *v) { /* :20-26 */
    v->curV++;
    if (v->curV == 0) {
        memset(v->mass, 0, sizeof(uint16_t) * v->n);
        v->curV++;
    }
}

/* ---------------------------------------------------------------- A1 */
/* One query.  base: row-major f32, row stride `stride` floats, scored over `dim`
 * floats (the padded dimension_, index_bipartite.cpp:2337).  Graph as CSR.
 * Returns 0, or -1 when fewer than k results exist (reference throws, :2408-2412). */
static int search_one(const float *base, uint64_t stride, uint32_t dim, int metric, const uint64_t *offsets,
                      const uint32_t *nbrs, uint32_t ep, const float *query, uint32_t k, uint32_t L,
                      oracle_visited *vl, oracle_pq *q, uint32_t *ids, float *dists, uint32_t *cmps_out,
                      uint32_t *hops_out, uint32_t *scanned_out) {
    q->size = 0; q->cur = 0; q->capacity = L;
    visited_reset(vl); /* getFreeVisitedList, :2333-2335 */
    uint16_t tag = vl->curV;
    uint16_t *mass = vl->mass;
    /* entry point scored and inserted, NOT marked visited (:2337-2353, :2349 commented out) */
    oracle_pq_insert(q, ep, oracle_dist(metric, base + (uint64_t)ep * stride, query, dim));
    uint32_t cmps = 0, hops = 0, scanned = 0;
    while (q->cur < q->size) { /* :2356 */
        oracle_nbr cur = oracle_pq_closest_unexpanded(q);
        const uint32_t *cn = nbrs + offsets[cur.id];
        uint32_t deg = (uint32_t)(offsets[cur.id + 1] - offsets[cur.id]);
        ++hops;
        scanned += deg;
        for (uint32_t j = 0; j < deg; ++j) { /* :2368 */
            uint32_t nb = cn[j];
            if (mass[nb] != tag) { /* :2378 */
                mass[nb] = tag;    /* :2385 */
                float d = oracle_dist(metric, base + (uint64_t)nb * stride, query, dim);
                ++cmps;
                oracle_pq_insert(q, nb, d); /* :2397-2398 */
            }
        }
    }
    *cmps_out = cmps; *hops_out = hops;
    if (scanned_out) *scanned_out = scanned;
    if (q->size < k) return -1;
    for (uint32_t i = 0; i < k; ++i) { /* :2414-2418 */
        ids[i] = q->data[i].id;
        dists[i] = q->data[i].distance;
    }
    return 0;
}

/* Batch of queries; loop shape of tests/test_search_roargraph.cpp:202-209
 * (omp parallel for schedule(dynamic,1), one pooled visited list per thread).
 * threads <= 0: all cores.  Returns number of queries that came up short of k. */
int oracle_search_batch(const float *base, uint64_t n, uint64_t stride, uint32_t dim, int metric,
                        const uint64_t *offsets, const uint32_t *nbrs, uint32_t ep, const float *queries,
                        uint64_t qstride, uint64_t nq, uint32_t k, uint32_t L, uint32_t *ids, float *dists,
                        uint32_t *cmps, uint32_t *hops, uint32_t *scanned, int threads) {
    int short_count = 0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_num_procs();
#else
    threads = 1;
#endif
    /* one visited list per thread, kept across calls like the reference's pool
     * (InitVisitedListPool once per index, outside the timed loop) */
    static oracle_visited *g_vl = NULL;
    static int g_threads = 0;
    static uint64_t g_n = 0;
    if (!g_vl || g_threads != threads || g_n != n) {
        for (int t = 0; g_vl && t < g_threads; ++t) free(g_vl[t].mass);
        free(g_vl);
        g_vl = (oracle_visited *)calloc((size_t)threads, sizeof(oracle_visited));
        for (int t = 0; t < threads; ++t) visited_init(&g_vl[t], n);
        g_threads = threads;
        g_n = n;
    }
#pragma omp parallel num_threads(threads) reduction(+ : short_count)
    {
#ifdef _OPENMP
        oracle_visited *vlp = &g_vl[omp_get_thread_num()];
#else
        oracle_visited *vlp = &g_vl[0];
#endif
        oracle_pq q;
        oracle_pq_init(&q, L);
#pragma omp for schedule(dynamic, 1)
        for (int64_t i = 0; i < (int64_t)nq; ++i) {
            uint32_t sc = 0;
            int rc = search_one(base, stride, dim, metric, offsets, nbrs, ep, queries + (uint64_t)i * qstride, k, L,
                                vlp, &q, ids + (uint64_t)i * k, dists + (uint64_t)i * k, &cmps[i], &hops[i], &sc);
            if (scanned) scanned[i] = sc;
            if (rc) short_count++;
        }
        oracle_pq_free(&q);
    }
    return short_count;
}

/* ---------------------------------------------------------------- F4: bipartite two-hop variant */
/* SearchBipartiteGraph, src/index_bipartite.cpp:282-356.  The reference draws its 10 entry points from
 * std::random_device (:287-294); here they are an argument so that a run can be replayed on the GPU.
 * Graph: CSR over base nodes [0, n_base) and training-query nodes [n_base, n_total). */
static int search_bipartite_one(const float *base, uint64_t stride, uint32_t dim, int metric, const uint64_t *offsets,
                                const uint32_t *nbrs, const uint32_t *init_ids, uint32_t n_init, const float *query,
                                uint32_t k, uint32_t L, oracle_visited *vl, oracle_pq *q, uint32_t *ids, float *dists,
                                uint32_t *cmps_out, uint32_t *hops_out) {
    q->size = 0; q->cur = 0; q->capacity = L;
    visited_reset(vl);
    uint16_t tag = vl->curV;
    uint16_t *mass = vl->mass;
    for (uint32_t i = 0; i < n_init; ++i) { /* :304-316: scored, inserted, marked */
        uint32_t id = init_ids[i];
        oracle_pq_insert(q, id, oracle_dist(metric, base + (uint64_t)id * stride, query, dim));
        mass[id] = tag;
    }
    uint32_t cmps = 0, hops = 0;
    while (q->cur < q->size) { /* :320 */
        oracle_nbr cur = oracle_pq_closest_unexpanded(q);
        hops++; /* :323, before the empty-row test :324-326 */
        for (uint64_t a = offsets[cur.id]; a < offsets[cur.id + 1]; ++a) {     /* :327 */
            uint32_t nbr = nbrs[a];
            for (uint64_t b = offsets[nbr]; b < offsets[nbr + 1]; ++b) {       /* :329 */
                uint32_t ns = nbrs[b];
                if (mass[ns] == tag) continue;                                 /* :330-332 */
                mass[ns] = tag;                                                /* :334 */
                float d = oracle_dist(metric, base + (uint64_t)ns * stride, query, dim);
                ++cmps;                                                        /* :340 */
                oracle_pq_insert(q, ns, d);
            }
        }
    }
    *cmps_out = cmps; *hops_out = hops;
    if (q->size < k) return -1; /* :347-351 */
    for (uint32_t i = 0; i < k; ++i) {
        ids[i] = q->data[i].id;
        dists[i] = q->data[i].distance;
    }
    return 0;
}

/* loop shape of tests/test_search_bipartite.cpp:194-201 */
int oracle_search_bipartite_batch(const float *base, uint64_t n_base, uint64_t stride, uint32_t dim, int metric,
                                  const uint64_t *offsets, const uint32_t *nbrs, const float *queries, uint64_t qstride,
                                  uint64_t nq, uint32_t k, uint32_t L, const uint32_t *init_ids, uint32_t n_init,
                                  uint32_t *ids, float *dists, uint32_t *cmps, uint32_t *hops, int threads) {
    int short_count = 0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_num_procs();
#else
    threads = 1;
#endif
#pragma omp parallel num_threads(threads) reduction(+ : short_count)
    {
        oracle_visited vl;
        visited_init(&vl, n_base); /* only base ids are ever marked (:315, :334) */
        memset(vl.mass, 0, sizeof(uint16_t) * (n_base ? n_base : 1));
        oracle_pq q;
        oracle_pq_init(&q, L);
#pragma omp for schedule(dynamic, 1)
        for (int64_t i = 0; i < (int64_t)nq; ++i) {
            int rc = search_bipartite_one(base, stride, dim, metric, offsets, nbrs, init_ids + (uint64_t)i * n_init,
                                          n_init, queries + (uint64_t)i * qstride, k, L, &vl, &q,
                                          ids + (uint64_t)i * k, dists + (uint64_t)i * k, &cmps[i], &hops[i]);
            if (rc) short_count++;
        }
        oracle_pq_free(&q);
        free(vl.mass);
    }
    return short_count;
}

/* qbaseNNbipartite, src/index_bipartite.cpp:235-280, sequential (sq ascending; the reference's OpenMP
 * loop appends to the target rows in lock order).  offsets has n_base + n_learn + 1 entries, nbrs room
 * for n_learn * min(K, M_pjbp) ids. */
void oracle_build_bipartite(const uint32_t *learn_knn, uint32_t n_learn, uint32_t K, uint32_t n_base, uint32_t M_pjbp,
                            uint64_t *offsets, uint32_t *nbrs) {
    uint32_t keep = K < M_pjbp ? K : M_pjbp; /* :256-259 */
    uint64_t nt = (uint64_t)n_base + n_learn;
    uint32_t *deg = (uint32_t *)calloc(nt, sizeof(uint32_t));
    for (uint32_t sq = 0; sq < n_learn; ++sq) {
        const uint32_t *nn = learn_knn + (uint64_t)sq * K;
        uint32_t tgt = nn[0]; /* choose_tgt = 0, :260-262 */
        for (uint32_t i = 0; i < keep; ++i)
            if (nn[i] != tgt) deg[n_base + sq]++; /* :263-268 */
        deg[tgt]++;                               /* :269-272 */
    }
    offsets[0] = 0;
    for (uint64_t v = 0; v < nt; ++v) offsets[v + 1] = offsets[v] + deg[v];
    memset(deg, 0, nt * sizeof(uint32_t));
    for (uint32_t sq = 0; sq < n_learn; ++sq) {
        const uint32_t *nn = learn_knn + (uint64_t)sq * K;
        uint32_t tgt = nn[0];
        for (uint32_t i = 0; i < keep; ++i)
            if (nn[i] != tgt) nbrs[offsets[n_base + sq] + deg[n_base + sq]++] = nn[i];
        nbrs[offsets[tgt] + deg[tgt]++] = n_base + sq;
    }
    free(deg);
}

/* IndexBipartite::Save / Load, src/index_bipartite.cpp:2045-2071: u32 npts; npts x {u32 deg; u32 nbr[deg]} */
int oracle_bipartite_save(const char *path, uint32_t npts, const uint64_t *offsets, const uint32_t *nbrs) {
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    fwrite(&npts, 4, 1, f);
    for (uint32_t i = 0; i < npts; ++i) {
        uint32_t deg = (uint32_t)(offsets[i + 1] - offsets[i]);
        fwrite(&deg, 4, 1, f);
        fwrite(nbrs + offsets[i], 4, deg, f);
    }
    fclose(f);
    return 0;
}

int oracle_bipartite_load(const char *path, uint32_t *npts, uint64_t *nedges, uint64_t *offsets, uint32_t *nbrs) {
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    if (fread(npts, 4, 1, f) != 1) { fclose(f); return -2; }
    uint64_t tot = 0;
    for (uint32_t i = 0; i < *npts; ++i) {
        uint32_t deg;
        if (fread(&deg, 4, 1, f) != 1) { fclose(f); return -2; }
        if (offsets) offsets[i] = tot;
        if (nbrs) {
            if (fread(nbrs + tot, 4, deg, f) != deg) { fclose(f); return -2; }
        } else {
            fseek(f, (long)deg * 4, SEEK_CUR);
        }
        tot += deg;
    }
    if (offsets) offsets[*npts] = tot;
    *nedges = tot;
    fclose(f);
    return 0;
}

/* ComputeRderr, tests/test_search_bipartite.cpp:38-62: mean over queries of the mean over the k ranks of
 * |r - g| / g with r the returned distance turned back into the ground truth's convention
 * (INNER_PRODUCT: -d; COSINE: 2 (1 + d)), float temporaries as there. */
double oracle_rderr(const float *gt_dist, uint32_t gt_dim, const float *res_dists, uint32_t nq, uint32_t k, int metric) {
    double total_err = 0;
    for (uint32_t i = 0; i < nq; ++i) {
        double err = 0.0;
        for (uint32_t j = 0; j < k; ++j) {
            float r = res_dists[(uint64_t)i * k + j];
            if (metric == 1) r = (float)(-1.0 * r);                       /* :46-49 */
            else if (metric == 4) r = (float)(2.0 * (1.0 - (-1.0 * r)));  /* :50-53 */
            float g = gt_dist[(uint64_t)i * gt_dim + j];
            err += fabsf(r - g) / (double)g;                              /* :57 */
        }
        total_err += err / (double)k;
    }
    return total_err / (double)nq;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_num_procs();
#else
    return 1;
#endif
}

/* ---------------------------------------------------------------- A8 */
/* util.h:214-225 (sequential float sum; the reference is built -Ofast so its own
 * summation order is compiler-chosen: compared within 1e-6 relative, not bitwise). */
void oracle_normalize(float *arr, uint64_t dim) {
    float sum = 0.0f;
    for (uint64_t i = 0; i < dim; ++i) sum += arr[i] * arr[i];
    sum = sqrtf(sum);
    for (uint64_t i = 0; i < dim; ++i) arr[i] = arr[i] / sum;
}

/* ---------------------------------------------------------------- A7 */
/* fbin: u32 n; u32 d; f32[n*d] (util.h:106-127).  Returns 0 / -1 open / -2 size. */
int oracle_fbin_meta(const char *path, uint32_t *n, uint32_t *d) {
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    if (fread(n, 4, 1, f) != 1 || fread(d, 4, 1, f) != 1) { fclose(f); return -2; }
    fseek(f, 0, SEEK_END);
    uint64_t fsize = (uint64_t)ftell(f);
    fclose(f);
    if (*d == 0) return -2;
    uint32_t contained = (uint32_t)((fsize - 8) / (*d) / 4);
    return contained == *n ? 0 : -2;
}

/* util.h:179-211: rows padded to a multiple of 8 floats, zero filled.  out has n*pad(d) floats. */
int oracle_fbin_load_padded(const char *path, uint32_t n, uint32_t d, float *out) {
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    uint64_t nd = ((uint64_t)d + 7) / 8 * 8;
    fseek(f, 8, SEEK_SET);
    for (uint64_t i = 0; i < n; ++i) {
        if (fread(out + i * nd, 4, d, f) != d) { fclose(f); return -2; }
        memset(out + i * nd + d, 0, (nd - d) * 4);
    }
    fclose(f);
    return 0;
}

/* graph index: u32 ep; u32 npts; npts x {u32 deg; u32 nbr[deg]} (index_bipartite.cpp:2097-2117).
 * Two-pass: first call with nbrs == NULL fills ep/npts/total edges. */
int oracle_graph_load(const char *path, uint32_t *ep, uint32_t *npts, uint64_t *nedges, uint64_t *offsets,
                      uint32_t *nbrs) {
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    if (fread(ep, 4, 1, f) != 1 || fread(npts, 4, 1, f) != 1) { fclose(f); return -2; }
    uint64_t tot = 0;
    for (uint32_t i = 0; i < *npts; ++i) {
        uint32_t deg;
        if (fread(&deg, 4, 1, f) != 1) { fclose(f); return -2; }
        if (offsets) offsets[i] = tot;
        if (nbrs) {
            if (fread(nbrs + tot, 4, deg, f) != deg) { fclose(f); return -2; }
        } else {
            fseek(f, (long)deg * 4, SEEK_CUR);
        }
        tot += deg;
    }
    if (offsets) offsets[*npts] = tot;
    *nedges = tot;
    fclose(f);
    return 0;
}

/* index_bipartite.cpp:2606-2619 */
int oracle_graph_save(const char *path, uint32_t ep, uint32_t npts, const uint64_t *offsets, const uint32_t *nbrs) {
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    fwrite(&ep, 4, 1, f);
    fwrite(&npts, 4, 1, f);
    for (uint32_t i = 0; i < npts; ++i) {
        uint32_t deg = (uint32_t)(offsets[i + 1] - offsets[i]);
        fwrite(&deg, 4, 1, f);
        fwrite(nbrs + offsets[i], 4, deg, f);
    }
    fclose(f);
    return 0;
}

/* ComputeRecall, tests/test_search_roargraph.cpp:23-36 */
float oracle_recall(uint32_t nq, uint32_t k, uint32_t gt_dim, const uint32_t *res, const uint32_t *gt) {
    uint32_t total = 0;
    for (uint32_t i = 0; i < nq; ++i)
        for (uint32_t a = 0; a < k; ++a) {
            uint32_t p = gt[(uint64_t)i * gt_dim + a];
            for (uint32_t b = 0; b < k; ++b)
                if (res[(uint64_t)i * k + b] == p) { total++; break; }
        }
    return (float)total / (float)(k * nq);
}
