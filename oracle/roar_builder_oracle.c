/*
 * roar_builder_oracle.c -- CPU restatement of the reference's RoarGraph BUILDER, used only as a
 * yardstick for graph quality ("parity = degree statistics + recall-vs-L curve").
 *
 * TEST INFRASTRUCTURE ONLY (same rule as roar_oracle.c, which this file is compiled with).
 *
 * Restated from /root/reference/src/index_bipartite.cpp (no source copied):
 *   BuildRoarGraph                              :143-233
 *   CalculateProjectionep                       :2004-2041
 *   LinkProjection                              :1043-1277
 *   PruneBiSearchBaseGetBase                    :1612-1694
 *   ProjectionAddReverse                        :1391-1432
 *   PruneProjectionReverseCandidates            :1527-1610
 *   SearchProjectionGraphInternal               :1279-1350
 *   PruneProjectionBaseSearchCandidates         :1846-1940
 *   SupplyAddReverse                            :1352-1389
 *   PruneProjectionInternalReverseCandidates    :1434-1525
 * The reference runs these phases under OpenMP with per-node mutexes, so its edge order depends on
 * thread timing; this restatement is the single-threaded order (training queries 0..n-1, nodes
 * 0..n-1).  It cannot be pinned against a reference build (index_bipartite.cpp needs Boost / robin-map
 * headers the image lacks): "parity unpinned" for this file, it is a reading of the algorithm.
 *
 * One quirk is reproduced on purpose: PruneProjectionInternalReverseCandidates builds its work list
 * with `std::vector<Neighbor> prune_queue(pruned_list.size())` and then push_backs, so the list starts
 * with size() zero-initialised entries (id 0, distance 0) (:1438).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

float oracle_dist(int metric, const float *a, const float *b, unsigned d); /* roar_oracle.c */

typedef struct { uint32_t id; float d; } bn; /* Neighbor, ordered by (distance, id) */
static int bn_less(const bn *a, const bn *b) { return a->d < b->d || (a->d == b->d && a->id < b->id); }
static int bn_cmp(const void *a, const void *b) {
    const bn *x = (const bn *)a, *y = (const bn *)b;
    return bn_less(x, y) ? -1 : (bn_less(y, x) ? 1 : 0);
}

typedef struct { uint32_t *v; uint32_t n, cap; } vec;
static void vpush(vec *a, uint32_t x) {
    if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 8; a->v = (uint32_t *)realloc(a->v, a->cap * 4); }
    a->v[a->n++] = x;
}
static int vhas(const vec *a, uint32_t x) {
    for (uint32_t i = 0; i < a->n; ++i) if (a->v[i] == x) return 1;
    return 0;
}
static void vset(vec *a, const uint32_t *src, uint32_t n) {
    a->n = 0;
    for (uint32_t i = 0; i < n; ++i) vpush(a, src[i]);
}

typedef struct {
    const float *base; uint64_t stride; uint32_t dim; int metric; uint32_t n, M;
    int flags; /* ablation switches for builder studies (0 = the reference's behaviour): bit 0 = a full row
                  drops a reverse edge instead of re-pruning; bit 1 = the connectivity pass prunes only the 63
                  nearest expanded nodes; bit 2 = a pivot's lists are appended (cap 2M) instead of overwritten;
                  bit 3 = no connectivity pass at all */
} ctx;
static float dd(const ctx *c, uint32_t a, uint32_t b) {
    return oracle_dist(c->metric, c->base + (uint64_t)a * c->stride, c->base + (uint64_t)b * c->stride, c->dim);
}

/* the occlusion loop shared by every prune variant: walk `q` (sorted) from index `from`, keep p unless
 * a kept r has dist(p, r) < dist(p, target) (:1640-1660 and its twins) */
static void occlusion_pass(const ctx *c, const bn *q, uint32_t nq, uint32_t from, uint32_t self, uint32_t *res,
                           uint32_t *nres) {
    for (uint32_t s = from; s < nq && *nres < c->M; ++s) {
        const bn *p = &q[s];
        int occl = 0;
        for (uint32_t t = 0; t < *nres; ++t) {
            if (p->id == res[t]) { occl = 1; break; }
            if (dd(c, p->id, res[t]) < p->d) { occl = 1; break; }
        }
        if (!occl && p->id != self) res[(*nres)++] = p->id;
    }
}

/* PruneBiSearchBaseGetBase (:1612-1694): pool = candidates with their distance to tgt */
static uint32_t prune_bisearch(const ctx *c, const bn *pool, uint32_t np, uint32_t tgt, uint32_t *res) {
    bn *bp = (bn *)malloc(sizeof(bn) * (np ? np : 1));
    uint32_t nb = 0, nres = 0;
    for (uint32_t i = 0; i < np; ++i) { /* unique ids, tgt excluded (:1622-1630) */
        if (pool[i].id == tgt) continue;
        int seen = 0;
        for (uint32_t j = 0; j < nb; ++j) if (bp[j].id == pool[i].id) { seen = 1; break; }
        if (!seen) bp[nb++] = pool[i];
    }
    if (nb == 0) { free(bp); return 0; }
    qsort(bp, nb, sizeof(bn), bn_cmp);
    res[nres++] = bp[0].id;
    occlusion_pass(c, bp, nb, 1, tgt, res, &nres);
    /* second pass over the unsorted pool (:1662-1683): same rule, can only re-reject */
    for (uint32_t s = 1; s < np && nres < c->M; ++s) {
        const bn *p = &pool[s];
        int in = 0, occl = 0;
        for (uint32_t t = 0; t < nres; ++t) if (res[t] == p->id) { in = 1; break; }
        if (in) continue;
        for (uint32_t t = 0; t < nres; ++t) if (dd(c, p->id, res[t]) < p->d) { occl = 1; break; }
        if (!occl && p->id != tgt) res[nres++] = p->id;
    }
    for (uint32_t i = 1; i < nb && nres < c->M; ++i) { /* back-fill (:1685-1691) */
        int in = 0;
        for (uint32_t t = 0; t < nres; ++t) if (res[t] == bp[i].id) { in = 1; break; }
        if (!in && bp[i].id != tgt) res[nres++] = bp[i].id;
    }
    free(bp);
    return nres;
}

/* PruneProjectionReverseCandidates (:1527-1610) / ...InternalReverse... (:1434-1525, `internal` = 1:
 * leading zero entries, no back-fill) */
static uint32_t prune_reverse(const ctx *c, uint32_t src, const uint32_t *list, uint32_t nl, int internal, uint32_t *res) {
    bn *q = (bn *)calloc((size_t)nl * 2 + 1, sizeof(bn));
    uint32_t nq = (internal && !(c->flags & 16)) ? nl : 0, nres = 0; /* :1438: nl zero-initialised entries first (flags bit 4: without them) */
    for (uint32_t i = 0; i < nl; ++i) {
        int seen = 0;
        for (uint32_t j = 0; j < nq; ++j) if (q[j].id == list[i]) { seen = 1; break; }
        if (!seen) { q[nq].id = list[i]; q[nq].d = dd(c, src, list[i]); nq++; }
    }
    qsort(q, nq, sizeof(bn), bn_cmp);
    uint32_t start = 0;
    if (q[start].id == src) start++;
    if (start < nq) {
        res[nres++] = q[start].id;
        occlusion_pass(c, q, nq, start + 1, src, res, &nres);
    }
    if (!internal)
        for (uint32_t i = 0; i < nl && nres < c->M; ++i) { /* back-fill in list order (:1602-1606) */
            int in = 0;
            for (uint32_t t = 0; t < nres; ++t) if (res[t] == list[i]) { in = 1; break; }
            if (!in) res[nres++] = list[i];
        }
    free(q);
    return nres;
}

/* ProjectionAddReverse (:1391-1432) with cap M / SupplyAddReverse (:1352-1389) with cap 2M */
static void add_reverse(const ctx *c, vec *g, uint32_t src, int supply) {
    const uint32_t cap = supply ? 2 * c->M : c->M;
    uint32_t *tmp = (uint32_t *)malloc(4 * (2 * c->M + 2) * 2);
    uint32_t *res = tmp + (2 * c->M + 2);
    for (uint32_t i = 0; i < g[src].n; ++i) {
        const uint32_t des = g[src].v[i];
        if (vhas(&g[des], src)) continue;
        if (g[des].n < cap) { vpush(&g[des], src); continue; }
        if (c->flags & 1) continue;
        uint32_t nl = 0;
        for (uint32_t j = 0; j < g[des].n; ++j) tmp[nl++] = g[des].v[j];
        tmp[nl++] = src;
        const uint32_t nr = prune_reverse(c, des, tmp, nl, supply, res);
        vset(&g[des], res, nr);
    }
    free(tmp);
}

/* sorted fixed-capacity queue of neighbor.h:138-223, restated in roar_oracle.c; a private copy here keeps
 * this file self-contained */
typedef struct { uint32_t id; float d; uint8_t f; } qe;
typedef struct { qe *e; size_t size, cap, cur; } pq;
static void pq_insert(pq *q, uint32_t id, float d) {
    qe nb = {id, d, 0};
    if (q->size == q->cap) {
        const qe *l = &q->e[q->size - 1];
        if (l->d < d || (l->d == d && l->id < id)) return;
    }
    size_t lo = 0, hi = q->size;
    while (lo < hi) {
        size_t mid = (lo + hi) >> 1;
        const qe *m = &q->e[mid];
        if (d < m->d || (d == m->d && id < m->id)) hi = mid;
        else if (m->id == id) return;
        else lo = mid + 1;
    }
    if (lo < q->cap) memmove(&q->e[lo + 1], &q->e[lo], (q->size - lo) * sizeof(qe));
    q->e[lo] = nb;
    if (q->size < q->cap) q->size++;
    if (lo < q->cur) q->cur = lo;
}

/* Returns 0.  learn_knn: [n_learn][K] nearest base ids of every training query, nearest first
 * (LoadLearnBaseKNN, :2622-2639).  Output CSR: offsets[n+1], nbrs (capacity n * 3 * M). */
int oracle_build_roargraph(const float *base, uint32_t n, uint64_t stride, uint32_t dim, int metric,
                           const uint32_t *learn_knn, uint32_t n_learn, uint32_t K, uint32_t M_sq, uint32_t M_pjbp,
                           uint32_t L_pjpq, uint32_t *ep_out, uint64_t *offsets, uint32_t *nbrs, int flags) {
    ctx c = {base, stride, dim, metric, n, M_pjbp, flags};
    const uint32_t M = M_pjbp;
    /* CalculateProjectionep (:2004-2041): node nearest (L2) the centroid */
    float *center = (float *)calloc(dim, sizeof(float));
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t d = 0; d < dim; ++d) center[d] += base[(uint64_t)i * stride + d];
    for (uint32_t d = 0; d < dim; ++d) center[d] /= (float)n;
    uint32_t ep = 0;
    float best = 0;
    for (uint32_t i = 0; i < n; ++i) {
        float diff = 0;
        for (uint32_t d = 0; d < dim; ++d) {
            const float t = center[d] - base[(uint64_t)i * stride + d];
            diff += t * t;
        }
        if (i == 0 || diff < best) { best = diff; ep = i; }
    }
    free(center);
    *ep_out = ep;

    vec *g = (vec *)calloc(n, sizeof(vec)), *sup = (vec *)calloc(n, sizeof(vec));
    bn *pool = (bn *)malloc(sizeof(bn) * (L_pjpq + K + 8));
    uint32_t *res = (uint32_t *)malloc(4 * (3 * M + 8));
    if (M_sq > K) M_sq = K;
    /* projection (:1059-1093) */
    for (uint32_t sq = 0; sq < n_learn; ++sq) {
        const uint32_t *nn = learn_knn + (uint64_t)sq * K;
        const uint32_t tgt = nn[0];
        uint32_t np = 0;
        for (uint32_t i = 0; i < M_sq; ++i) {
            if (nn[i] == tgt) continue;
            pool[np].id = nn[i];
            pool[np].d = dd(&c, nn[i], tgt);
            np++;
        }
        const uint32_t nr = prune_bisearch(&c, pool, np, tgt, res);
        if (flags & 4) {
            for (uint32_t i = 0; i < nr && g[tgt].n < 2 * M; ++i) if (!vhas(&g[tgt], res[i])) vpush(&g[tgt], res[i]);
        } else {
            vset(&g[tgt], res, nr);
        }
        add_reverse(&c, g, tgt, 0);
    }
    for (uint32_t v = 0; v < n; ++v) add_reverse(&c, g, v, 0); /* :1096-1099 */
    for (uint32_t v = 0; v < n; ++v) {                         /* :1102-1129 */
        if (g[v].n <= M) continue;
        uint32_t np = 0;
        for (uint32_t j = 0; j < g[v].n; ++j) {
            const uint32_t u = g[v].v[j];
            int seen = (u == v);
            for (uint32_t t = 0; t < np && !seen; ++t) seen = pool[t].id == u;
            if (seen) continue;
            pool[np].id = u;
            pool[np].d = dd(&c, u, v);
            np++;
        }
        const uint32_t nr = prune_bisearch(&c, pool, np, v, res);
        vset(&g[v], res, nr);
    }
    /* connectivity enhancement (:1183-1262) */
    for (uint32_t v = 0; v < n; ++v) vset(&sup[v], g[v].v, g[v].n);
    uint8_t *visited = (uint8_t *)malloc(n);
    pq q;
    q.e = (qe *)malloc(sizeof(qe) * (L_pjpq + 1));
    q.cap = L_pjpq;
    bn *ret = (bn *)malloc(sizeof(bn) * ((size_t)n + 1));
    for (uint32_t v = 0; v < n && !(flags & 8); ++v) {
        /* SearchProjectionGraphInternal (:1279-1350) over supply_nbrs_, from the entry point */
        memset(visited, 0, n);
        q.size = 0; q.cur = 0;
        const float *query = base + (uint64_t)v * stride;
        pq_insert(&q, ep, oracle_dist(metric, base + (uint64_t)ep * stride, query, dim));
        visited[ep] = 1;
        uint32_t nret = 0;
        while (q.cur < q.size) {
            q.e[q.cur].f = 1;
            const size_t pre = q.cur;
            while (q.cur < q.size && q.e[q.cur].f) q.cur++;
            const qe cur = q.e[pre];
            ret[nret].id = cur.id; ret[nret].d = cur.d; nret++;
            for (uint32_t j = 0; j < sup[cur.id].n; ++j) {
                const uint32_t nb = sup[cur.id].v[j];
                if (visited[nb] || nb == v) continue;
                visited[nb] = 1;
                pq_insert(&q, nb, oracle_dist(metric, base + (uint64_t)nb * stride, query, dim));
            }
        }
        uint32_t w = 0; /* drop the node itself (:1208-1213) */
        for (uint32_t i = 0; i < nret; ++i) if (ret[i].id != v) ret[w++] = ret[i];
        nret = w;
        /* PruneProjectionBaseSearchCandidates (:1846-1940) */
        uint32_t nres = 0;
        if (nret) {
            qsort(ret, nret, sizeof(bn), bn_cmp);
            if ((flags & 2) && nret > 63) nret = 63;
            uint32_t start = 0;
            if (ret[start].id == v) start++;
            while (start < nret && vhas(&g[v], ret[start].id)) ++start;
            if (start < nret) {
                res[nres++] = ret[start].id;
                occlusion_pass(&c, ret, nret, start + 1, v, res, &nres);
            }
        }
        vset(&sup[v], res, nres);
        add_reverse(&c, sup, v, 1);
    }
    for (uint32_t v = 0; v < n; ++v) { /* :1230-1258: supply rows over M are re-pruned */
        if (sup[v].n <= M) continue;
        uint32_t nret = 0;
        for (uint32_t j = 0; j < sup[v].n; ++j) {
            const uint32_t u = sup[v].v[j];
            int seen = 0;
            for (uint32_t t = 0; t < nret && !seen; ++t) seen = ret[t].id == u;
            if (seen) continue;
            ret[nret].id = u; ret[nret].d = dd(&c, u, v); nret++;
        }
        uint32_t nres = 0;
        qsort(ret, nret, sizeof(bn), bn_cmp);
        uint32_t start = 0;
        if (nret && ret[start].id == v) start++;
        while (start < nret && vhas(&g[v], ret[start].id)) ++start;
        if (start < nret) {
            res[nres++] = ret[start].id;
            occlusion_pass(&c, ret, nret, start + 1, v, res, &nres);
        }
        vset(&sup[v], res, nres);
    }
    /* merge up to 2M supply edges (:1262-1275) */
    uint64_t tot = 0;
    for (uint32_t v = 0; v < n; ++v) {
        uint32_t added = 0;
        const uint32_t g0 = g[v].n;
        for (uint32_t j = 0; j < sup[v].n && added < 2 * M; ++j) {
            int in = 0;
            for (uint32_t t = 0; t < g0; ++t) if (g[v].v[t] == sup[v].v[j]) { in = 1; break; }
            if (!in) { vpush(&g[v], sup[v].v[j]); added++; }
        }
        offsets[v] = tot;
        memcpy(nbrs + tot, g[v].v, (size_t)g[v].n * 4);
        tot += g[v].n;
    }
    offsets[n] = tot;
    for (uint32_t v = 0; v < n; ++v) { free(g[v].v); free(sup[v].v); }
    free(g); free(sup); free(pool); free(res); free(visited); free(q.e); free(ret);
    return 0;
}

/* PruneBiSearchBaseGetBase alone, for unit tests of the device pruning kernel: cand = candidate ids (the
 * target excluded or not), returns the kept ids in order. */
uint32_t oracle_prune_bisearch(const float *base, uint64_t stride, uint32_t dim, int metric, uint32_t tgt,
                               const uint32_t *cand, uint32_t ncand, uint32_t M, uint32_t *res) {
    ctx c = {base, stride, dim, metric, 0, M, 0};
    bn *pool = (bn *)malloc(sizeof(bn) * (ncand + 1));
    uint32_t np = 0;
    for (uint32_t i = 0; i < ncand; ++i) {
        if (cand[i] == tgt) continue;
        pool[np].id = cand[i];
        pool[np].d = dd(&c, cand[i], tgt);
        np++;
    }
    const uint32_t nr = prune_bisearch(&c, pool, np, tgt, res);
    free(pool);
    return nr;
}
