// roar_plan.cpp -- launch plans of the SearchRoarGraph kernels: LDS carve-up, kernel form, grid (plan_search), and the calibration that measures what no option fixes on the loaded index (calibrate_plan, resolve_plan)
// (round 5: roar_capi.cpp was one 3 203-line translation unit; it is now cut by concern.  Shared declarations: roar_internal.h.)
#include "roar_internal.h"

using namespace roar_impl;

namespace roar_impl {

// LDS carve-up and grid for one search launch.  `pick`: measured choices for what no option fixes (null: defaults).
int plan_search(roar_index *idx, uint64_t nq, uint32_t k, uint32_t L, LaunchPlan *plan, bool bipartite, const PlanPick *pick) {
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
    const bool stream = pick && pick->stream && !bipartite && !idx->building;  // batch-stream form (roar_stream.cpp): one wave per query, forgetful table
    if (!stream && !bipartite && !idx->building && L > 64 && idx->hub_mode != 1 && (idx->hub_mode == 2 || (dense && L >= 100))) {
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
    if (stream) {
        // workgroups of several launches live side by side: nothing may be indexed by workgroup number in HBM (tag lists, spill lists)
        if (!lds_fits) return fail(ROAR_ERR_ARG, "the batch-stream form runs the on-chip tag table, which %llu base points do not fit", (unsigned long long)idx->n);
        if (mode == ROAR_VISITED_EXACT || mode == ROAR_VISITED_LDS)
            return fail(ROAR_ERR_ARG, "the batch-stream form runs the forgetful on-chip tag table: option visited_mode must be AUTO or LDS_FORGETFUL");
        mode = ROAR_VISITED_LDS_FORGETFUL;
    } else if (mode == ROAR_VISITED_AUTO) {
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
    uint32_t smult = 1;  // sets = smult << sets_log2
    if (mode == ROAR_VISITED_LDS) {
        uint32_t want = (p.screen || idx->dim_pad * 4u < 768u) ? 4096u : 8192u;
        if (pick && pick->entries > 0 && !idx->building && !bipartite) want = pick->entries;
        if (idx->vis_entries > 0) want = (uint32_t)idx->vis_entries;
        uint32_t min_sets = (uint32_t)((idx->n + 65533) / 65534);
        uint32_t sets = std::max(std::max(want / 4, min_sets), 16u);
        while ((1u << sets_log2) < sets) sets_log2++;
        // 5 * 2^k and 3 * 2^k sets (1 280, 1 536, 2 560, 3 072 ... tags) where that is what was asked for: the search kernel only.  The
        // smallest of 5/8, 3/4 and 1 times the power of two that holds what was asked for.
        if (!bipartite && sets_log2 >= 3 && sets <= (5u << (sets_log2 - 3)) && (5u << (sets_log2 - 3)) >= min_sets) {
            sets_log2 -= 3;
            smult = 5;
        } else if (!bipartite && sets_log2 >= 2 && sets <= (3u << (sets_log2 - 2)) && (3u << (sets_log2 - 2)) >= min_sets) {
            sets_log2 -= 2;
            smult = 3;
        }
        entries = (4u * smult) << sets_log2;
        if ((uint64_t)entries * 2 > 128 * 1024)
            return fail(ROAR_ERR_ARG, "visited table of %u entries does not fit in LDS", entries);
    }
    p.vis_sets_log2 = sets_log2 | (smult == 3 ? 0x80000000u : 0u) | (smult == 5 ? 0x10000000u : 0u);  // (+ bit 30 once the spill lists are known to fit, below)
    uint32_t off = 0;
    p.off_pool = off;
    off += round_up(8 * (L + 1), 16);
    // candidate ids / keys of one adjacency round trip: three 64-word chunks of a row -- two when no row has a third (stride <= 128 words:
    // the benchmark graphs), one for stride 64.  768 bytes less per query is a workgroup more per CU at every other L_pq of a long pool
    // (the LDS is handed out in 1 280-byte blocks, below), and long pools live on resident workgroups: T2I-10M, L_pq = 1 000, 2 048 tags,
    // nine per CU 36.2 ms -> ten per CU (profiles/r04_long_pools_lds.txt).
    const uint32_t trip = (use_hub || bipartite || idx->adj_stride > 128u) ? 192u : idx->adj_stride > 64u ? 128u : 64u;
    // One wave per query (never the multi-wave or the bipartite forms: their waves append to ck while others still read cid): the ids sit
    // in the UPPER HALF of the keys' block.  A scoring pass reads its ids before it writes its keys (one wave: LDS operations in program
    // order), the keys of the c rows scored so far end at byte 8 c, the ids not yet scored start at byte 4 trip + 4 c >= 8 c for every
    // c <= trip.  512 bytes less per query at two chunks per row: with the 1 280-tag table (below) a 12th workgroup per CU at L_pq = 1 000.
    const bool maybe_multiwave = !stream && !idx->building && (idx->search_kernel >= 2 || (idx->search_kernel == 0 && dense && nq <= (uint64_t)idx->n_cus));
    const bool alias_ids = !bipartite && !use_hub && !maybe_multiwave && !idx->building;
    p.off_ck = off;
    off += bp_wg ? 8 * 64 * 8 : 8 * trip;   // eight waves x 64 keys in the workgroup form; else one adjacency round trip
    if (alias_ids) {
        p.off_cid = p.off_ck + 4 * trip;
    } else {
        p.off_cid = off;
        off += bp_wg ? 4 * 64 * 8 : 4 * trip;
    }
    p.off_q = off;
    off += round_up(4 * idx->dim_pad, 16);
    if (p.screen) off += round_up(idx->dim_pad, 16);  // the query's 8-bit codes (screen8i_pass)
    p.off_vis = off;
    off += 2 * entries;
    // "edge floats" (RoarSearchParams::adj_meta): the screened neighbours' floats travel with the adjacency row -- inner product, code rows
    // of whole lines (128-d, 512-d), the one-wave search form without hub bitmap / phase timers / counters
    bool ann = false;
    if (p.screen && !bipartite && !idx->building && !use_hub && mode == ROAR_VISITED_LDS && idx->ip && roar_screen8_split(idx->dim_pad) && idx->edge_floats_mode != 1 &&
        !idx->profile && !(idx->traffic_dev && nq <= idx->traffic_cap) && idx->search_kernel <= 1 && idx->max_deg <= 127u) {
        ensure_edge_floats(idx);
        ann = idx->adj_meta != nullptr;
    }
    if (ann) {
        p.adj_meta = idx->adj_meta;
        p.off_cmeta = off;
        off += 8 * trip;
    }
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
    if (stream) {  // the workgroup's copy of the slot it works on + its place in the stream (roar_search_v2.h stream_take)
        p.off_stream = off;
        off += 128;
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
    sel.stream = stream ? 1 : 0;
    sel.ann = ann ? 1 : 0;
    idx->last_ann = sel.ann;
    if (stream && sel.prof) return fail(ROAR_ERR_ARG, "profile_phases: the batch-stream form has no instrumented kernel");
    if (sel.prof && !(idx->ip && idx->dim_pad == 200) && !(!idx->ip && idx->dim_pad == 128))
        return fail(ROAR_ERR_ARG, "profile_phases: the instrumented kernels exist for the 200-d inner-product and the 128-d L2 shapes only");
    int occ = 0;
    auto occupancy = [&](int *out) -> int {
        // (asked once per kernel form and LDS size: a stream of launches of one shape must not pay a runtime query each)
        const uint64_t key = ((uint64_t)p.lds_bytes << 32) | ((uint64_t)(sel.stream & 1) << 31) | ((uint64_t)(sel.ann & 1) << 15) | ((uint64_t)(sel.form & 7) << 28) | ((uint64_t)(sel.waves & 7) << 24) |
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
    } else if (idx->search_kernel != 1 && !idx->building && !stream) {
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
    idx->last_wpc = wpc;
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
        const uint64_t sets = (uint64_t)smult << sets_log2;
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
}  // namespace roar_impl

namespace roar_impl {

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
    uint32_t lband = 0;  // half-octave bands of L_pq: [2^(b/2), 2^((b+1)/2)) ...
    while (lband < 40 && std::pow(2.0, 0.5 * (lband + 1)) <= (double)L) lband++;
    // ... up to L_pq = 512; beyond, one band per 1 280-byte LDS block of the pool (160 entries): a long pool's table is chosen against
    // what the pool leaves of the CU's LDS, and a pick made at L_pq = 1 500 (3 072 tags, eight workgroups per CU) ran the launches at
    // 1 600-1 900 of the same half-octave at seven and six per CU -- 3-12 % behind the smaller table round 5 had picked there
    // (profiles/r06_sweep_t2i10m.csv, first pass)
    if (L >= 512u) lband = 32u + std::min<uint32_t>(200u, round_up(8u * (L + 1u), 16u) / 1280u);
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
    // ... and by WHICH choices were free when the entry was measured: what is measured with an option fixed (screen forced on, the visited
    // form given, a table size given) is conditional on it and says nothing about the launch with that option back on AUTO
    const uint64_t fixed = (uint64_t)(idx->screen_mode & 3) | ((uint64_t)(idx->visited_mode & 3) << 2) | ((uint64_t)(idx->vis_entries & 0xFFFFF) << 4);
    // ... and by the REGIME the entry is for: a launch that has the chip to itself, or one that shares it (a grid cap, a stream's
    // workgroups).  The candidates are timed the same way -- alone, on the whole chip -- but ranked differently (calibrate_plan: a
    // shared chip lives on resident workgroups, and the timings of one launch at a time cannot see that)
    const uint64_t shared = idx->max_grid > 0 ? 1u : 0u;
    return (uint64_t)lband | ((uint64_t)band << 8) | ((uint64_t)(idx->waves_per_cu & 0xFF) << 20) | (fixed << 28) | (shared << 52);
}

const roar_index::Calibration *find_calibration(const roar_index *idx, uint64_t key) {
    for (const auto &c : idx->plan_cache)
        if (c.key == key) return &c;
    return nullptr;
}


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
    bool used_lists = false;  // some candidate used the per-workgroup lists in HBM: later launches of other streams are ordered behind it
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
            // a candidate that uses the per-workgroup lists in HBM (HBM tags, spill lists) must not run beside a launch of another stream
            // that still uses them: the ordering search_device applies (roar_launch.cpp)
            if (pl.p.tags != nullptr || idx->last_used_tags) {
                if (idx->ev_valid && idx->last_stream != st && hipStreamWaitEvent(st, idx->ev1, 0) != hipSuccess) return -1.f;
                if (pl.p.tags != nullptr) used_lists = true;
            }
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
        uint32_t grid = 0;  // workgroups of the timed launch (the sample is at least one residency: how many the plan keeps resident)
        uint32_t wpc = 0;   // ... per CU, as the LDS and the registers allow (before the grid is cut to the sample)
    };
    std::vector<Cand> tried;
    tried.reserve(16);
    auto try_pick = [&](int screen, int mode, uint32_t entries, int protect = -1) -> const Cand * {
        PlanPick pk;
        pk.screen = screen, pk.mode = mode, pk.entries = entries, pk.protect = protect;
        Cand c{pk, -1.f, 0, 0, 0, protect};
        c.ms = time_pick(pk, &c.entries, &c.mode, &c.screen);
        c.grid = (uint32_t)idx->last_grid;
        c.wpc = (uint32_t)idx->last_wpc;
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
            tried.push_back(Cand{pk, default_ms, eu, mu, su, -1, (uint32_t)idx->last_grid, (uint32_t)idx->last_wpc});
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
        // (the descent follows the best ON-CHIP table, whether or not the HBM tags lead at this point -- they are the default plan from
        // L_pq = 1 536 on, and a 4 096-tag table a little behind them said nothing about the 2 048-tag one: T2I-10M, L_pq = 1 600, HBM tags
        // 58-64 ms run to run, 4 096 tags 59.9, 2 560 tags 49.1 -- the sweep's run kept the HBM tags, the probe's the table)
        auto best_table = [&]() -> const Cand * {
            const Cand *b = nullptr;
            for (const Cand &c : tried)
                if (c.mode == ROAR_VISITED_LDS && (!b || c.ms < b->ms)) b = &c;
            return b;
        };
        if (!hard_error && entries_free) {
            const Cand *b = best_table();
            if (b) {
                const int sc = screen_free ? b->screen : -1;
                // (a smaller table buys resident workgroups and nothing else: where the whole sample was resident with the best table so
                // far -- a launch of a few dozen queries -- there is nothing to buy, only noise to pick from and rows to forget)
                if (b->entries <= 4096 && b->grid < S) {
                    for (uint32_t e = 2048; e >= 1024 && !hard_error; e >>= 1) {
                        const Cand *c = try_pick(sc, ROAR_VISITED_LDS, e);
                        b = best_table();
                        if (!c || !b || b->entries != c->entries) break;  // (did not win: stop here)
                    }
                } else if (b->entries >= 8192) {
                    try_pick(sc, ROAR_VISITED_LDS, 16384);
                }
                b = best_table();
                if (!hard_error && b && L >= 256 && b->entries >= 1024 && b->entries <= 4096 && b->grid < S) {
                    const uint32_t e0 = b->entries;
                    try_pick(sc, ROAR_VISITED_LDS, e0 + e0 / 2);
                    if (!hard_error && (e0 & (e0 - 1)) == 0) try_pick(sc, ROAR_VISITED_LDS, e0 - e0 / 4);
                    // the 5 * 2^k sizes on both sides of the winner (1 280 between 1 024 and 1 536 ...): at a long pool 256 tags are a workgroup per CU
                    b = best_table();
                    if (!hard_error && b) {
                        const uint32_t eb = b->entries;
                        uint32_t lo5 = 640;
                        while (lo5 * 2 < eb) lo5 *= 2;  // the largest 5 * 2^k below the winner; twice that is the smallest above it
                        if (lo5 >= 1024) try_pick(sc, ROAR_VISITED_LDS, lo5);
                        if (!hard_error && lo5 * 2 != eb && lo5 * 2 <= 5120) try_pick(sc, ROAR_VISITED_LDS, lo5 * 2);
                    }
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
                if (plan_search(idx, S, k, L, &pl, false, &pk) == ROAR_OK) {
                    // (only where the other setting is another launch: one wave per query, pool in LDS, no hub bitmap, tags under 2^15)
                    const int with = idx->last_tag_protect;
                    PlanPick other = pk;
                    other.protect = with ? 0 : 1;
                    LaunchPlan pl2;
                    if (pl.sel.form == ROAR_FORM_SEARCH && pl.sel.waves == 1 && !pl.sel.regpool && !pl.sel.hub &&
                        plan_search(idx, S, k, L, &pl2, false, &other) == ROAR_OK && idx->last_tag_protect != with)
                        try_pick(pk.screen, pk.mode, pk.entries, with ? 0 : 1);
                } else g_err = keep;
            }
        }
    }
    if (used_lists && hipEventRecord(idx->ev1, st) == hipSuccess) {
        idx->ev_valid = true;
        idx->last_stream = st;
        idx->last_used_tags = true;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(sc_ids);
    idx->max_grid = cap_keep;
    if (hard_error) return hard_error;
    // (a launch that shares the chip does not want the per-workgroup lists in HBM: launches that use them run one behind the other, and a
    // stream cannot use them at all -- an on-chip table wins there whenever one ran)
    const bool shared = cap_keep > 0;
    bool any_table = false;
    for (const Cand &c : tried) any_table = any_table || c.mode != ROAR_VISITED_EXACT;
    const Cand *best = nullptr;
    for (const Cand &c : tried) {
        if (shared && any_table && c.mode == ROAR_VISITED_EXACT) continue;
        if (!best || c.ms < best->ms) best = &c;
    }
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
    // Among the candidates within 0.5 % of the fastest (noise): HBM tags first (they return the reference's cmps exactly); among on-chip
    // tables the one that keeps MORE WORKGROUPS RESIDENT, then the bigger one (it forgets less: cmps >= the reference's; ids / distances /
    // hops never depend on any of this), then the default plan.  One launch at a time they are a tie by definition; launches in flight
    // and the batch stream live on resident workgroups.  (Rounds 4-5 let the bigger table win the tie: once the kernel-argument reads
    // stopped costing a round trip per hop, the 8 192-tag table -- eight workgroups per CU instead of twelve -- came within 0.3-1.2 % of
    // the 4 096-tag one on the sample in some runs, and the batch stream of those runs was 3 % slower: 7.15 against 7.39 M QPS.)
    // A launch that SHARES the chip (grid cap, stream: plan_key's regime bit) widens the tie to 2 %: one launch at a time the tables from
    // 2 048 to 8 192 tags are within +-1.5 % of each other on the benchmark graph, run to run, while through a stream the 8 192-tag table
    // (eight workgroups per CU) is 3 % slower than the 4 096-tag one (twelve) for 10 000-query batches and 4.7 % slower for 1 250-query ones
    // (profiles/r06_table_regime_probe.txt) -- the driver's round-5 run picked it on a 0.6 % "win" and lost 3 % of `value`.
    {
        // (a sample that was resident as a whole with every candidate -- a small launch, timed in a fraction of a millisecond -- cannot
        // tell the tables apart by residency, and its timings scatter by percents: the tie is 5 % wide there, and the bigger table --
        // it forgets less -- takes it)
        bool all_resident = true;
        for (const Cand &c : tried) all_resident = all_resident && c.grid >= S;
        const float lim = best->ms * (all_resident ? 1.05f : shared ? 1.02f : 1.005f);
        auto fidelity = [&](const Cand &c) -> uint64_t {
            return ((uint64_t)(c.mode == ROAR_VISITED_EXACT ? 1 : 0) << 60) | ((uint64_t)(c.grid & 0xFFFFFu) << 36) | ((uint64_t)c.entries << 8) |
                   (uint64_t)(&c == &tried.front() ? 1 : 0);
        };
        for (const Cand &c : tried) {
            if (shared && any_table && c.mode == ROAR_VISITED_EXACT) continue;
            if (c.ms <= lim && c.screen == best->screen && fidelity(c) > fidelity(*best)) best = &c;
        }
    }
    out->screen = screen_free ? best->screen : -1;
    out->mode = mode_free ? best->mode : -1;
    out->entries = (entries_free && best->mode == ROAR_VISITED_LDS) ? best->entries : 0;
    out->protect = idx->tag_protect < 0 ? best->protect : -1;
    *calibrated = true;
    try {
        roar_index::Calibration c{key, out->screen, out->mode, out->entries, out->protect, best->ms, default_ms, S, L, runs,
                                  (best->mode == ROAR_VISITED_LDS && out->entries) ? best->wpc : 0u};
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
uint64_t stream_calibration_min_queries(const roar_index *idx) { return (uint64_t)std::max(idx->n_cus, 1) * 12u; }

// A cached pick, applied at THIS L_pq: the table it names was chosen against the LDS the pool left at the L_pq it was measured at; a
// longer pool of the same band may leave room for one workgroup per CU less with it (a band spans a 1 280-byte block of pool: at most
// one residency step).  A long pool lives on resident workgroups, so the table steps down -- 3/4, 5/8, 1/2 ... of itself, to 1 024
// tags or the smallest the point count allows -- until the plan keeps what the measurement had (T2I-10M: the pick of L_pq = 1 600,
// 2 560 tags at eight per CU, served 1 700 at seven: 3.7 % behind round 5; profiles/r06_sweep_t2i10m.csv).
int plan_with_cached_pick(roar_index *idx, uint64_t nq, uint32_t k, uint32_t L, LaunchPlan *plan, PlanPick pick, uint32_t want_wpc) {
    int e = plan_search(idx, nq, k, L, plan, false, &pick);
    if (e || want_wpc == 0 || pick.entries == 0) return e;
    static const uint32_t steps[] = {16384, 12288, 10240, 8192, 6144, 5120, 4096, 3072, 2560, 2048, 1536, 1280, 1024};
    while ((uint32_t)idx->last_wpc < want_wpc && idx->last_mode == ROAR_VISITED_LDS) {
        const uint32_t have = (uint32_t)idx->last_vis_entries;
        uint32_t next = 0;
        for (uint32_t st : steps)
            if (st < have) {
                next = st;
                break;
            }
        if (next == 0) break;
        pick.entries = next;
        LaunchPlan smaller;
        if ((e = plan_search(idx, nq, k, L, &smaller, false, &pick))) return e;
        if ((uint32_t)idx->last_vis_entries >= have) {  // (the point count allows nothing smaller: the plan before it stands)
            pick.entries = have;
            return plan_search(idx, nq, k, L, plan, false, &pick);
        }
        *plan = smaller;
    }
    return ROAR_OK;
}

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
        return plan_with_cached_pick(idx, nq, k, L, plan, pick, L > c->L ? c->wpc : 0u);
    } else if (idx->max_grid > 0 && nq < stream_calibration_min_queries(idx)) {
        // a launch under a grid cap takes the calibration of the largest batches (plan_key) -- which a sample of less than one
        // residency of the chip cannot make: defaults for this call, nothing cached
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

// *provisional (may be null): set when the plan stands on defaults only because this batch is too small a sample to measure on -- the
// stream asks again when a larger batch arrives (roar_stream.cpp).
int resolve_stream_plan(roar_index *idx, const float *qdev, uint64_t nq, uint32_t qstride, uint32_t k, uint32_t L, hipStream_t st, LaunchPlan *plan,
                        bool *provisional) {
    PlanPick pick;
    uint32_t want_wpc = 0;
    idx->last_calibrated = 0;
    if (provisional) *provisional = false;
    const bool sane = idx->base && idx->adj && idx->gn == idx->n && L != 0 && k != 0 && k <= L;  // (else: plan_search words the error)
    if (sane && calibratable(idx)) {
        // A stream's workgroups share the chip with each other whatever the size of the batches: the regime of the largest batches,
        // as for a launch under a grid cap (plan_key) -- its calibration is taken, or made now on this batch's queries.
        const int cap_keep = idx->max_grid;
        idx->max_grid = 0x7FFFFFFF;
        const uint64_t key = plan_key(idx, nq, L);
        int e = ROAR_OK;
        if (const roar_index::Calibration *c = find_calibration(idx, key)) {
            pick.screen = c->screen, pick.mode = c->mode, pick.entries = c->entries, pick.protect = c->protect;
            idx->last_calibrated = 1;
            if (L > c->L) want_wpc = c->wpc;
        } else if (nq < stream_calibration_min_queries(idx)) {
            // as resolve_plan does for a launch under a grid cap: a sample of less than one residency of the chip cannot make the calibration
            // of the largest batches (every table ties, noise picks -- and the pick would be cached for every later stream and capped launch
            // of this L_pq band): defaults for now, nothing cached
            if (provisional) *provisional = true;
        } else if (qdev) {
            bool done = false;
            e = calibrate_plan(idx, qdev, nq, qstride, k, L, st, &pick, &done);
            idx->last_calibrated = done ? 1 : 0;
        }
        idx->max_grid = cap_keep;
        if (e) return e;
    }
    if (pick.mode == ROAR_VISITED_EXACT) pick.mode = -1, pick.entries = 0;  // (HBM tags won the calibration: the stream runs the table anyway)
    pick.stream = 1;
    return plan_with_cached_pick(idx, 1ull << 30, k, L, plan, pick, want_wpc);
}
}  // namespace roar_impl
