// pipeline.cpp -- host orchestration of the whole hot path behind mauve_align():
//   doAlignment (mauveAligner.cpp:70): MaskedMemHash multi-MUMs (:523-531,585) -> MultiplicityFilter /
//   EliminateOverlaps (:596-600) -> Aligner::align (:698): LCBs by greedy breakpoint elimination with
//   default weight 3*w*N (:648-653), recursive anchoring of gaps > min_recursive_gap_length
//   (:127,670-672), gapped alignment of every inter-anchor interval through the GappedAligner seam
//   (:674-676) -> addUnalignedIntervals (:748) -> IntervalList (WriteStandardAlignment :746-760).
// Device work: seed pass (seed_pass.hip) and batched DP (dp_batch.hip).  Host work: the sequential
// chaining over the compact LCB graph and the assembly of SoA results.  No oracle code is used.
#include "common.hpp"
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace {

inline uint8_t base_at(const uint64_t *w, int64_t i) { return (uint8_t)((w[(size_t)(i >> 5)] >> (2 * (i & 31))) & 3); }

// inter-anchor interval of genome g between anchors a (left in genome-0 order) and b, LCB orientation
inline void gap_of(const int64_t *a, const int64_t *b, int g, int64_t &lo, int64_t &len, bool &rev)
{
    const int64_t sa = a[1 + g], sb = b[1 + g];
    int64_t hi;
    if (sa > 0) { lo = sa + a[0]; hi = sb - 1; rev = false; }
    else { lo = -sb + b[0]; hi = -sa - 1; rev = true; }
    len = hi - lo + 1; if (len < 0) len = 0;
}

}  // namespace

int recursive_anchoring(mauve_ctx *c, const mauve_params *p, int w0, std::vector<MatchVec> &chains, int N, const int *gmap);

// LCB extension (lcb_extension, mauveAligner.cpp:95; SetMaxExtensionIterations :687-690), frozen replacement
// up to max_extension_iters rounds, each with a seed two weights lighter than the last.  A round
// searches N-way MEMs only where no LCB lies -- one masked seed pass over the resident genomes, the mask being the
// union of the LCB extents -- adds what it finds to the surviving anchors, eliminates overlaps and recomputes the
// LCBs with the same minimum weight; it is kept only if it raises the number of anchored columns.
static int extend_lcbs(mauve_ctx *c, const mauve_params *p, int w, int64_t lcbw, MatchVec &m, std::vector<int64_t> &match_lcb,
                       int64_t &nl)
{
    const int N = m.N;
    const uint32_t full = N >= 32 ? 0xffffffffu : ((1u << N) - 1);
    int w_e = w;
    MatchVec base(N); ChainOrders base_ord; bool have_base = false;         // the surviving anchors and their per-genome orders
    for (int iter = 0; iter < p->max_extension_iters; iter++) {
        w_e -= 2;
        if (w_e < 5) break;
        const uint64_t pe = mauve_get_seed(w_e, 0);
        if (!pe) break;
        const int64_t span_e = mauve_seed_length(pe);
        static const bool trace = getenv("MAUVE_TRACE") != nullptr;
        const double tr0 = now_ms();
        // LCB extents
        std::vector<int64_t> lo((size_t)nl * N, 0), hi((size_t)nl * N, 0);
        int64_t cols_before = 0;
        for (size_t i = 0; i < m.size(); i++) {
            const int64_t l = match_lcb[i]; if (l < 0) continue;
            cols_before += m.len(i);
            for (int g = 0; g < N; g++) {
                const int64_t a = std::llabs(m.st(i)[g]), b = a + m.len(i) - 1;
                int64_t &L = lo[(size_t)l * N + g], &H = hi[(size_t)l * N + g];
                if (L == 0 || a < L) L = a;
                if (b > H) H = b;
            }
        }
        // mask bitmap: every base masked, then the valid pieces (gaps between extents, at least one seed long) cleared
        GenomeSet gs = main_genome_set(c);
        gs.mask_off.assign((size_t)N, 0);
        size_t words = 0;
        for (int g = 0; g < N; g++) { gs.mask_off[(size_t)g] = words; words += (size_t)((c->lens[g] + 63) / 64) + 2; }
        HIPCHK(c, c->pin_mask.ensure(words * 8));
        uint64_t *bits = c->pin_mask.as<uint64_t>();
        memset(bits, 0xff, words * 8);
        bool starved = false;
        std::vector<std::pair<int64_t, int64_t>> sp((size_t)nl);
        for (int g = 0; g < N && !starved; g++) {
            for (int64_t l = 0; l < nl; l++) sp[(size_t)l] = {lo[(size_t)l * N + g], hi[(size_t)l * N + g]};
            std::sort(sp.begin(), sp.end());
            uint64_t *M = bits + gs.mask_off[(size_t)g];
            int64_t cur = 1; bool any = false;
            for (int64_t l = 0; l <= nl; l++) {
                const int64_t vlo = cur, vhi = l < nl ? sp[(size_t)l].first - 1 : c->lens[g];
                if (vhi - vlo + 1 >= span_e) {
                    any = true;
                    for (int64_t b = vlo - 1; b < vhi;) {               // clear [vlo-1, vhi) word-wise
                        const int64_t wd = b >> 6, e = std::min<int64_t>(vhi, (wd + 1) << 6);
                        const int n = (int)(e - b), sh = (int)(b & 63);
                        M[wd] &= ~((n == 64 ? ~0ULL : ((1ULL << n) - 1ULL)) << sh);
                        b = e;
                    }
                }
                if (l < nl && sp[(size_t)l].second + 1 > cur) cur = sp[(size_t)l].second + 1;
            }
            if (!any) starved = true;
        }
        if (starved) break;
        if (c->has_invalid) for (size_t k = 0; k < words && k < c->h_invalid.size(); k++) bits[k] |= c->h_invalid[k];     // same word layout
        HIPCHK(c, c->placed_mask.ensure(words * 8));
        HIPCHK(c, hipMemcpyAsync(c->placed_mask.p, bits, words * 8, hipMemcpyHostToDevice, c->stream));
        gs.vmask = &c->placed_mask;          // (the seed pass syncs the stream before the staging buffer is touched again)
        int64_t nx = 0;
        const double tr1 = now_ms();
        int rc = seedpass_run(c, gs, pe, MAUVE_MODE_MEM, full, 1, nullptr, 0, &nx);
        if (rc) return rc;
        if (trace) fprintf(stderr, "[trace] lcb extension round %d: mask %.3f ms, seed pass %.3f ms, %lld new matches\n", iter, tr1 - tr0, now_ms() - tr1, (long long)nx);
        if (nx == 0) continue;
        const double tr2 = now_ms();
        // survivors + new matches, canonical order (both lists already are: merge)
        auto less = [N](const int64_t *a, const int64_t *b) {                // N-way records: |start0|, starts, length
            const int64_t sa = std::llabs(a[1]), sb = std::llabs(b[1]);
            if (sa != sb) return sa < sb;
            for (int g = 0; g < N; g++) if (a[1 + g] != b[1 + g]) return a[1 + g] < b[1 + g];
            return a[0] < b[0];
        };
        MatchVec ext(N); ext.resize((size_t)nx);
        for (int64_t i = 0; i < nx; i++) {
            ext.len((size_t)i) = c->match_len[(size_t)i];
            std::copy(&c->match_start[(size_t)i * N], &c->match_start[(size_t)i * N] + N, ext.st((size_t)i));
        }
        // The new matches lie outside every LCB extent in every genome, the surviving anchors inside: a new match can
        // only overlap another new one.  Overlap elimination of the merged list is therefore the elimination among the
        // new matches alone (a handful) next to the untouched anchors, and the per-genome orders of the merged list
        // are the anchors' orders with the new matches slipped in -- no sort of the whole list per round.
        if (!have_base) {
            base.d.clear();
            for (size_t i = 0; i < m.size(); i++) if (match_lcb[i] >= 0) base.push(m.rec(i));
            // Crops of the overlap elimination can carry a match past a neighbour (dense, overlapping lists: seed families):
            // the merged list of a round is in canonical order of the CURRENT records, so the anchors are put in that order first
            {
                std::vector<size_t> idx(base.size());
                for (size_t i = 0; i < idx.size(); i++) idx[i] = i;
                auto by = [&](size_t x, size_t y) { return less(base.rec(x), base.rec(y)); };
                if (!std::is_sorted(idx.begin(), idx.end(), by)) {
                    std::stable_sort(idx.begin(), idx.end(), by);
                    MatchVec t(N); t.reserve(base.size());
                    for (size_t i : idx) t.push(base.rec(i));
                    base.d.swap(t.d);
                }
            }
            host_left_orders(base, base_ord);
            have_base = true;
        }
        host_eliminate_overlaps(ext);
        const size_t nb = base.size(), ne = ext.size();
        // merged list: the few new records go in at their places (binary search), the anchors move in blocks
        const size_t R1 = (size_t)(1 + N);
        std::vector<size_t> ins(ne);                                         // new record b goes before anchor ins[b]
        for (size_t b = 0; b < ne; b++) {
            size_t lo2 = b ? ins[b - 1] : 0, hi2 = nb;
            while (lo2 < hi2) { const size_t mid = (lo2 + hi2) / 2; if (less(ext.rec(b), base.rec(mid))) hi2 = mid; else lo2 = mid + 1; }
            ins[b] = lo2;
        }
        MatchVec comb(N); comb.d.resize((nb + ne) * R1);
        std::vector<uint32_t> at_b(nb), at_e(ne);                            // where the records went
        for (size_t b = 0, a = 0; b <= ne; b++) {
            const size_t a_end = b < ne ? ins[b] : nb;
            std::copy(base.d.begin() + (std::ptrdiff_t)(a * R1), base.d.begin() + (std::ptrdiff_t)(a_end * R1), comb.d.begin() + (std::ptrdiff_t)((a + b) * R1));
            for (size_t x = a; x < a_end; x++) at_b[x] = (uint32_t)(x + b);
            a = a_end;
            if (b < ne) { at_e[b] = (uint32_t)(a + b); std::copy(ext.rec(b), ext.rec(b) + R1, comb.d.begin() + (std::ptrdiff_t)((a + b) * R1)); }
        }
        ChainOrders orders; orders.ord.resize((size_t)N);
        std::vector<uint32_t> eo(ne);
        for (int g = 0; g < N; g++) {
            for (size_t k = 0; k < ne; k++) eo[k] = (uint32_t)k;
            std::sort(eo.begin(), eo.end(), [&](uint32_t x, uint32_t y) { return std::llabs(ext.st(x)[g]) < std::llabs(ext.st(y)[g]); });
            const std::vector<uint32_t> &bo = base_ord.ord[(size_t)g];
            std::vector<uint32_t> &o = orders.ord[(size_t)g];
            o.resize(nb + ne);
            size_t a = 0;
            for (size_t b = 0; b <= ne; b++) {
                size_t a_end = nb;
                if (b < ne) {                                                // first anchor (in this genome's order) right of the new match
                    const int64_t key = std::llabs(ext.st(eo[b])[g]);
                    size_t lo2 = a, hi2 = nb;
                    while (lo2 < hi2) { const size_t mid = (lo2 + hi2) / 2; if (std::llabs(base.st(bo[mid])[g]) < key) lo2 = mid + 1; else hi2 = mid; }
                    a_end = lo2;
                }
                for (size_t x = a; x < a_end; x++) o[x + b] = at_b[bo[x]];
                a = a_end;
                if (b < ne) o[a + b] = at_e[eo[b]];
            }
        }
        std::vector<int64_t> ml2; int64_t nl2 = 0;
        const double tr3 = now_ms();
        host_lcb_chain(comb, lcbw, p->collinear != 0, ml2, nl2, &orders);
        if (trace) fprintf(stderr, "[trace] lcb extension round %d: merge %.3f ms, lcb %.3f ms\n", iter, tr3 - tr2, now_ms() - tr3);
        int64_t cols_after = 0;
        for (size_t i = 0; i < comb.size(); i++) if (ml2[i] >= 0) cols_after += comb.len(i);
        if (cols_after > cols_before) {
            // the anchors of the next round: what survived this one, orders thinned and renumbered accordingly
            bool all_alive = true;
            for (size_t i = 0; i < comb.size() && all_alive; i++) all_alive = ml2[i] >= 0;
            if (all_alive) { base.d = comb.d; base_ord.ord.swap(orders.ord); }
            else {
                std::vector<uint32_t> renum(comb.size(), 0xffffffffu);
                base.d.clear();
                for (size_t i = 0; i < comb.size(); i++) if (ml2[i] >= 0) { renum[i] = (uint32_t)base.size(); base.push(comb.rec(i)); }
                for (int g = 0; g < N; g++) {
                    std::vector<uint32_t> &bo = base_ord.ord[(size_t)g];
                    bo.clear();
                    for (uint32_t x : orders.ord[(size_t)g]) if (renum[x] != 0xffffffffu) bo.push_back(renum[x]);
                }
            }
            m.d.swap(comb.d); match_lcb.swap(ml2); nl = nl2;
        }
    }
    return MAUVE_OK;
}

// The searches of the seed family of weight w (ranks 0..2, longest seed first; ties: the higher rank, as the call site's
// sort leaves them, progressiveMauve.cpp:515-521) merged into one list (host_merge_matches).
int seed_family_matches(mauve_ctx *c, const GenomeSet &gs, int w, int mode, uint64_t mask, MatchVec &out)
{
    const int n = gs.nseq;
    int order[3] = {0, 1, 2};
    std::sort(order, order + 3, [&](int a, int b) {
        const int la = mauve_seed_length(mauve_get_seed(w, a)), lb = mauve_seed_length(mauve_get_seed(w, b));
        return la != lb ? la > lb : a > b;
    });
    out = MatchVec(n);
    for (int k = 0; k < 3; k++) {
        const uint64_t pat = mauve_get_seed(w, order[k]);
        if (!pat) continue;
        int64_t nm = 0;
        const int rc = seedpass_run(c, gs, pat, mode, mask, 1, nullptr, 0, &nm);
        if (rc) return rc;
        static thread_local MatchVec cur_keep(1);                  // (kept from call to call: no fresh megabytes and their page faults per search)
        MatchVec &cur = cur_keep; cur.N = n; cur.d.clear(); cur.resize((size_t)nm);
        for (int64_t i = 0; i < nm; i++) {
            cur.len((size_t)i) = c->match_len[(size_t)i];
            std::copy(&c->match_start[(size_t)i * n], &c->match_start[(size_t)i * n] + n, cur.st((size_t)i));
        }
        const double tm0 = now_ms();
        if (k == 0) out.d.swap(cur.d); else host_merge_matches(out, cur);
        static const bool trace = getenv("MAUVE_TRACE") != nullptr;
        if (trace) fprintf(stderr, "[trace] seed family: seed %d gave %lld matches, merge %.3f ms\n", order[k], (long long)nm, now_ms() - tm0);
    }
    return MAUVE_OK;
}

static const bool g_trace_pipeline = getenv("MAUVE_TRACE") != nullptr;     // read once, not in the timed path

// ---- the whole path in three phases, so that the DP intervals of one alignment can be sharded over ranks ----
// begin : seed pass, chaining, recursive anchoring, interval descriptors        (every rank, deterministic)
// dp    : gapped alignment of a subset of the intervals                         (each rank its share)
// finish: assembly of the interval table from the columns of ALL intervals      (every rank)
static int align_begin(mauve_ctx *c, const mauve_params *p, bool device_front = false, const MatchVec *given = nullptr, bool want_tail = false)
{
    const int N = c->nseq;
    AlignState &S = c->ast;
    S.reset();
    c->rec_flags.clear();
    S.p = *p; S.N = N; S.t0 = now_ms();
    AlignResult &R = c->res;
    // keep the capacity of the result vectors across calls (a fresh 20 MB column buffer per call costs more in
    // page faults than the whole seed pass)
    R.sz = mauve_align_sizes();
    R.mum_length.clear(); R.mum_start.clear(); R.lcb_left.clear(); R.lcb_right.clear(); R.lcb_weight.clear();
    R.anchor_length.clear(); R.anchor_start.clear(); R.anchor_lcb.clear(); R.iv_left.clear(); R.iv_right.clear();
    R.iv_reverse.clear(); R.col_off.clear(); R.n_cols = 0; R.dp_score.clear();
    R.dev_pending = false; R.cols_pending = false; R.stale = false; R.genomes_replaced = false; R.dev_na = 0; R.dev_nm = 0; R.cols_ext = nullptr;
    memset(&c->stage, 0, sizeof c->stage);

    int64_t sum = 0; for (int g = 0; g < N; g++) sum += c->lens[g];
    S.sum = sum;
    int w = p->seed_weight > 0 ? p->seed_weight : mauve_default_seed_weight(sum / N);
    uint64_t pat = p->seed_pattern ? p->seed_pattern : mauve_get_seed(w, p->seed_rank);
    if (!pat) { c->err = "align: no seed pattern for this weight/rank"; return MAUVE_ERR_ARG; }
    w = mauve_seed_weight(pat);
    const uint32_t full = N >= 32 ? 0xffffffffu : ((1u << N) - 1);
    S.full = full;

    // ---- seed pass: N-way multi-MUMs (the multiplicity filter is pushed into the join) ----
    // In its shadow (the host would otherwise wait for the kernels): bring the column buffer back to the
    // "all anchors" state, i.e. undo the gap ranges the previous call wrote (see align_finish).
    static const bool no_shadow = getenv("MAUVE_NO_SHADOW") != nullptr;        // A/B switch
    if (R.cols_fill == full && !R.cols_dirty.empty() && !no_shadow)
        c->shadow = [c, full]() {
            AlignResult &Rr = c->res;
            for (const auto &d : Rr.cols_dirty) std::fill(Rr.cols.begin() + d.first, Rr.cols.begin() + d.first + d.second, full);
            Rr.cols_dirty.clear();
        };
    int64_t nm = 0;
    int rc = MAUVE_OK;
    // LCB extension without taking the chains off the device (extend_dev.hip): plain genomes, length-weighted LCBs.  MAUVE_HOST_EXTEND: A/B switch
    static const bool host_extend = getenv("MAUVE_HOST_EXTEND") != nullptr;
    const bool do_extend = p->extend_lcbs && p->lcb_scoring == MAUVE_LCB_SCORE_LENGTH;      // (score-weighted LCBs are not extended: the rule counts columns)
    // (collinear: the greedy step runs down to ONE node, so an old LCB can lose against the new matches -- the unit-level re-chaining
    //  of extend_dev.hip assumes old LCBs survive; that rare option takes the host rounds)
    const bool ext_on_device = do_extend && !host_extend && !c->has_invalid && !c->has_contigs && !p->collinear;
    bool chains_ready = false;                       // S.chains filled from the device anchors (extension done there, recursion to follow on the host)
    MatchVec family(N);
    if (!given && p->seed_family) {
        // DESIGN.md S3b (progressiveMauve.cpp:502-546): one search per seed of the family, longest seed first, merged like the
        // matches of one finder; the merged list then stands where a caller's list would
        if (p->seed_pattern) { c->err = "align: seed_family takes its patterns from the weight, not from seed_pattern"; return MAUVE_ERR_ARG; }
        const double tf0 = now_ms();
        rc = seed_family_matches(c, main_genome_set(c), w, p->mode, full, family);
        if (rc) return rc;
        if (g_trace_pipeline) fprintf(stderr, "[trace] seed family: three searches + merges %.3f ms, %zu matches\n", now_ms() - tf0, family.size());
        given = &family;
    }
    if (given) {
        // the caller's list: its N-way matches in canonical order (|start 0|, starts, length) stand where the seed pass's stood
        c->shadow = nullptr;
        if (R.cols_fill == full && !R.cols_dirty.empty()) {
            for (const auto &d : R.cols_dirty) std::fill(R.cols.begin() + d.first, R.cols.begin() + d.first + d.second, full);
            R.cols_dirty.clear();
        }
        std::vector<size_t> keep;
        for (size_t i = 0; i < given->size(); i++) {
            bool nway = given->len(i) > 0;
            for (int g = 0; g < N && nway; g++) nway = given->st(i)[g] != 0;
            if (nway) keep.push_back(i);
        }
        auto canon = [&](size_t a, size_t b) {
            const int64_t *x = given->rec(a), *y = given->rec(b);
            const int64_t sa = std::llabs(x[1]), sb = std::llabs(y[1]);
            if (sa != sb) return sa < sb;
            for (int g = 0; g < N; g++) if (x[1 + g] != y[1 + g]) return x[1 + g] < y[1 + g];
            if (x[0] != y[0]) return x[0] < y[0];
            return a < b;
        };
        if (!std::is_sorted(keep.begin(), keep.end(), canon)) std::sort(keep.begin(), keep.end(), canon);     // a merged family list comes in order
        nm = (int64_t)keep.size();
        c->match_len.resize((size_t)nm); c->match_start.resize((size_t)nm * N);
        for (int64_t i = 0; i < nm; i++) {
            c->match_len[(size_t)i] = given->len(keep[(size_t)i]);
            std::copy(given->st(keep[(size_t)i]), given->st(keep[(size_t)i]) + N, &c->match_start[(size_t)i * N]);
        }
        c->n_matches = nm; c->dev_rec_n = -1;
        // A long list goes where the seed pass would have left its own: into sorted_rec in the device layout (int64 length[n], start[n * N]),
        // so that overlap elimination, LCBs, extension and the tail run on the device as they do after a search (chain_dev.hip; a
        // list the device chain declines -- MAUVE_ERR_LIMIT -- falls back to the host chain below).  Same threshold as the seed pass's
        // device sort; MAUVE_GIVEN_ON_HOST: A/B switch.
        static const int64_t dev_min = getenv("MAUVE_CANON_DEVICE_MIN") ? atol(getenv("MAUVE_CANON_DEVICE_MIN")) : 16384;
        static const bool given_on_host = getenv("MAUVE_GIVEN_ON_HOST") != nullptr;
        if (!given_on_host && nm >= dev_min && nm < (1LL << 31) && p->lcb_scoring == MAUVE_LCB_SCORE_LENGTH) {
            const size_t rb = (size_t)nm * (1 + (size_t)N) * 8;
            HIPCHK(c, c->sorted_rec.ensure(rb + 64));
            HIPCHK(c, c->pin_tab.ensure(rb + 64));
            int64_t *hp = c->pin_tab.as<int64_t>();
            memcpy(hp, c->match_len.data(), (size_t)nm * 8);
            memcpy(hp + nm, c->match_start.data(), (size_t)nm * N * 8);
            HIPCHK(c, hipMemcpyAsync(c->sorted_rec.p, hp, rb, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            c->dev_rec_n = nm; c->match_nseq = N;
        }
    } else {
        static const bool host_tail_env = getenv("MAUVE_HOST_TAIL") != nullptr;
        c->lazy_matches_ok = want_tail && !host_tail_env && (!do_extend || ext_on_device);     // the list may stay in HBM (device tail below)
        rc = seedpass_run(c, main_genome_set(c), pat, p->mode, full, 1, nullptr, 0, &nm);
        c->lazy_matches_ok = false;
        c->shadow = nullptr;
        if (rc) return rc;
    }
    S.nm = nm;
    if (!c->matches_pending) { R.mum_length = c->match_len; R.mum_start = c->match_start; }
    const double t1 = now_ms();
    c->stage.seed_ms = t1 - S.t0;

    // ---- chaining ----
    // On the device when the seed pass left the list there in canonical order (chain_dev.hip); on the host otherwise
    // (small lists, ties in the canonical order) or when the device repair gives up.  MAUVE_HOST_CHAIN: A/B switch.
    MatchVec &m = S.m; m.N = N;
    const int64_t lcbw = p->lcb_weight >= 0 ? p->lcb_weight : (int64_t)3 * w * N;
    std::vector<int64_t> &match_lcb = S.match_lcb; int64_t nl = 0;
    static const bool host_chain = getenv("MAUVE_HOST_CHAIN") != nullptr;
    if (p->lcb_scoring != MAUVE_LCB_SCORE_LENGTH && p->lcb_scoring != MAUVE_LCB_SCORE_SP) { c->err = "align: unknown lcb_scoring"; return MAUVE_ERR_ARG; }
    bool on_device = !host_chain && nm > 0 && c->dev_rec_n == nm && p->lcb_scoring == MAUVE_LCB_SCORE_LENGTH;   // score weights: host chain
    double t1b = t1;
    if (on_device) {
        rc = chain_device_core(c, N, lcbw, p->collinear != 0, nl);
        if (rc == MAUVE_ERR_LIMIT) on_device = false;
        else if (rc) return rc;
        else {
            // The chains can stay on the device when nothing on the host has to look at them: no LCB extension, and no
            // inter-anchor gap long enough for the recursion (counted on the device).  The DP front end and the assembly
            // then run there too (mauve_align), and the host sees only the per-LCB rows.
            static const bool host_tail = getenv("MAUVE_HOST_TAIL") != nullptr;      // A/B switch
            if (want_tail && !host_tail && (!do_extend || ext_on_device) && nl > 0) {
                int64_t na = 0, nrec = 0;
                rc = chain_order_device(c, N, nl, p->min_recursive_gap, &na, &nrec);
                if (rc) return rc;
                const uint32_t cap = (uint32_t)c->dev_rec_n;
                S.dv_len = c->ch_anch.as<int32_t>(); S.dv_st = S.dv_len + cap; S.dv_lcb = S.dv_st + (size_t)cap * N;
                bool ext_declined = false;
                if (do_extend && na >= 1) {
                    std::vector<int64_t> lw;
                    rc = extend_lcbs_device(c, p, w, lcbw, N, &S.dv_len, &S.dv_st, &S.dv_lcb, &na, &nl, &nrec, lw);
                    // MAUVE_ERR_LIMIT: the unit-level rounds met a case they do not cover (an old LCB would die) and have put everything back: the
                    // chains go to the host as after any other device chain and the match-level rounds run there (extend_lcbs below)
                    if (rc == MAUVE_ERR_LIMIT) { ext_declined = true; rc = MAUVE_OK; c->err.clear(); }
                    else if (rc) return rc;
                    else { R.lcb_weight.swap(lw); S.lw_from_host = true; }
                }
                if (ext_declined) { S.dv_len = S.dv_st = S.dv_lcb = nullptr; }
                else if (na >= 2 && (!p->recursive || nrec == 0)) {
                    S.nl = nl; S.n_anchor = na; S.dev_tail = true;
                    const double t2 = now_ms();
                    c->stage.chain_ms = t2 - t1;
                    if (g_trace_pipeline) fprintf(stderr, "[trace] chain: on the device %.3f ms, %lld anchors in %lld LCBs stay there\n", t2 - t1, (long long)na, (long long)nl);
                    S.t_dp0 = t2;
                    S.open = true;
                    return MAUVE_OK;
                }
                if (do_extend && !ext_declined) {
                    // the extension is done; what follows (recursion) wants the chains on the host: the anchors come over as they are
                    const size_t rb = (size_t)na * (2 + (size_t)N) * 4;
                    HIPCHK(c, c->pin_chain.ensure(256 + rb + (size_t)na + 64));
                    int32_t *hl = reinterpret_cast<int32_t *>(c->pin_chain.as<char>() + 256), *hs = hl + na, *hb = hs + (size_t)na * N;
                    uint8_t *hf = reinterpret_cast<uint8_t *>(hb + na);
                    HIPCHK(c, hipMemcpyAsync(hl, S.dv_len, (size_t)na * 4, hipMemcpyDeviceToHost, c->stream));
                    HIPCHK(c, hipMemcpyAsync(hs, S.dv_st, (size_t)na * N * 4, hipMemcpyDeviceToHost, c->stream));
                    HIPCHK(c, hipMemcpyAsync(hb, S.dv_lcb, (size_t)na * 4, hipMemcpyDeviceToHost, c->stream));
                    if (p->recursive) {                  // which gaps the recursion will look at: flagged on the device, the work list visits only those
                        HIPCHK(c, c->ext_work.ensure((size_t)na + 64));
                        rc = rec_gap_flags_device(c, S.dv_len, S.dv_st, S.dv_lcb, na, N, p->min_recursive_gap, c->ext_work.as<uint8_t>());
                        if (rc) return rc;
                        HIPCHK(c, hipMemcpyAsync(hf, c->ext_work.p, (size_t)na, hipMemcpyDeviceToHost, c->stream));
                    }
                    HIPCHK(c, hipStreamSynchronize(c->stream));
                    if (p->recursive) c->rec_flags.assign(hf, hf + na); else c->rec_flags.clear();
                    std::vector<MatchVec> &chs = S.chains;
                    chs.resize((size_t)nl);
                    for (auto &ch : chs) { ch.N = N; ch.d.clear(); }
                    // (chain order: LCB by LCB, so every chain is one contiguous stretch of the list -- sized once, filled in place)
                    for (int64_t a = 0; a < na;) {
                        int64_t e = a + 1;
                        while (e < na && hb[e] == hb[a]) e++;
                        MatchVec &ch = chs[(size_t)hb[a]];
                        const size_t at = ch.d.size();
                        ch.d.resize(at + (size_t)(e - a) * (1 + (size_t)N));
                        int64_t *o = ch.d.data() + at;
                        for (int64_t q = a; q < e; q++) { *o++ = hl[q]; for (int g = 0; g < N; g++) *o++ = hs[(size_t)q * N + g]; }
                        a = e;
                    }
                    chains_ready = true;
                    if (g_trace_pipeline) fprintf(stderr, "[trace] chain (device): %lld extended anchors to the host (recursion to follow)\n", (long long)na);
                    S.dv_len = S.dv_st = S.dv_lcb = nullptr; S.lw_from_host = false;
                }
            }
            static const bool no_keep = getenv("MAUVE_NO_KEEP_MUMS") != nullptr;      // A/B switch
            if (c->matches_pending && chains_ready && !no_keep && c->dev_rec_n == c->n_matches) {
                // the chains came over as anchors; nothing on the host reads the match list before the caller fetches it: it stays in HBM,
                // set aside while the recursion's passes use the seed workspace (back in place for the assembly: assemble_device)
                std::swap(c->sorted_rec, c->sorted_rec_keep);
                S.mums_kept = c->n_matches;
                c->matches_pending = false; c->dev_rec_n = -1;
            } else if (c->matches_pending) {             // the host goes on: it needs its copy of the list after all
                rc = seed_matches_to_host(c);
                if (rc) return rc;
                R.mum_length = c->match_len; R.mum_start = c->match_start;
            }
            if (!chains_ready) {
                rc = chain_device_copy_back(c, N, m, match_lcb);
                if (rc) return rc;
            }
        }
        t1b = now_ms();
    }
    if (c->matches_pending) {                            // device chain gave up: host chain on the host copy
        rc = seed_matches_to_host(c);
        if (rc) return rc;
        R.mum_length = c->match_len; R.mum_start = c->match_start;
    }
    if (!on_device) {
        m.resize((size_t)nm);
        for (int64_t i = 0; i < nm; i++) {
            m.len((size_t)i) = c->match_len[(size_t)i];
            std::copy(&c->match_start[(size_t)i * N], &c->match_start[(size_t)i * N] + N, m.st((size_t)i));
        }
        ChainOrders orders;
        static const bool elim_compact = getenv("MAUVE_ELIM_COMPACT") != nullptr;       // A/B switch
        host_eliminate_overlaps(m, &orders, elim_compact);           // default: dead records stay in m, with lcb -1 below
        t1b = now_ms();
        if (p->lcb_scoring == MAUVE_LCB_SCORE_SP) {              // LCB weight = sum-of-pairs score of its anchors
            std::vector<int64_t> mw;
            rc = match_sp_scores(c, m, nullptr, &p->scoring, mw);
            if (rc) return rc;
            const int64_t minw = p->lcb_weight >= 0 ? p->lcb_weight : sp_default_min_weight(w, N, &p->scoring);
            host_lcb_chain(m, minw, p->collinear != 0, match_lcb, nl, &orders, mw.data());
            S.match_weight.swap(mw);
        } else
        host_lcb_chain(m, lcbw, p->collinear != 0, match_lcb, nl, &orders);
    }
    if (do_extend && !chains_ready) {
        rc = extend_lcbs(c, p, w, lcbw, m, match_lcb, nl);
        if (rc) return rc;
    }
    S.nl = nl;
    std::vector<MatchVec> &chains = S.chains;
    if (!chains_ready) {
        chains.resize((size_t)nl);                        // element buffers keep their capacity
        for (auto &ch : chains) { ch.N = N; ch.d.clear(); }
        R.lcb_weight.assign((size_t)nl, 0);
        for (size_t i = 0; i < m.size(); i++) {
            int64_t l = match_lcb[i]; if (l < 0) continue;
            chains[(size_t)l].push(m.rec(i));              // m is sorted by genome-0 start (canonical order)
            R.lcb_weight[(size_t)l] += S.match_weight.empty() ? m.len(i) * N : S.match_weight[i];
        }
    }
    const double t2 = now_ms();
    c->stage.chain_ms = t2 - t1;
    if (g_trace_pipeline) fprintf(stderr, "[trace] chain: eliminate_overlaps %.3f ms, lcb %.3f ms\n", t1b - t1, t2 - t1b);

    // ---- recursive anchoring ----
    if (p->recursive) {
        rc = recursive_anchoring(c, p, w, chains, N, nullptr);
        if (rc) return rc;
    }
    const double t3 = now_ms();
    c->stage.recurse_ms = t3 - t2;

    // ---- inter-anchor intervals.  Descriptors only: the bases are gathered from the resident packed genomes on
    // the device. ----
    S.gaps.reserve((size_t)R.mum_length.size() + 16);
    if (device_front) {              // mauve_align: the interval table is made on the device (dp_run_from_anchors)
        for (int64_t l = 0; l < nl; l++) {
            const MatchVec &ch = chains[(size_t)l];
            S.n_anchor += (int64_t)ch.size();
            for (size_t i = 0; i < ch.size(); i++) S.anchor_cols += ch.len(i);
        }
    } else
    for (int64_t l = 0; l < nl; l++) {
        const MatchVec &ch = chains[(size_t)l];
        S.n_anchor += (int64_t)ch.size();
        for (size_t i = 0; i < ch.size(); i++) {
            S.anchor_cols += ch.len(i);
            if (i + 1 == ch.size()) break;
            int64_t tot = 0, mx = 0; int nonempty = 0;
            int64_t lo[MAUVE_MAX_SEQ], ln[MAUVE_MAX_SEQ]; bool rv[MAUVE_MAX_SEQ];
            for (int g = 0; g < N; g++) {
                gap_of(ch.rec(i), ch.rec(i + 1), g, lo[g], ln[g], rv[g]);
                tot += ln[g]; mx = std::max(mx, ln[g]); nonempty += ln[g] > 0;
            }
            if (tot == 0) continue;
            AlignState::GapRef gr; gr.lcb = l; gr.idx = (int64_t)i; gr.dp = false; gr.dp_slot = -1; gr.tot = tot;
            if (p->gapped && nonempty >= 2 && mx <= dp_len_limit(p)) {
                gr.dp = true; gr.dp_slot = S.n_dp++;
                for (int g = 0; g < N; g++) {
                    DpSeqDesc d; d.genome = g; d.rev = rv[g]; d.lo0 = lo[g] - 1; d.len = ln[g];
                    S.desc.push_back(d);
                }
                S.code_total += tot;
            }
            S.gaps.push_back(gr);
        }
    }
    S.t_dp0 = now_ms();
    if (g_trace_pipeline) fprintf(stderr, "[trace] interval table: %.3f ms (%lld gaps, %lld dp)\n", S.t_dp0 - t3, (long long)S.gaps.size(), (long long)S.n_dp);
    S.open = true;
    return MAUVE_OK;
}

// DP of the intervals idx[0..n) (slots of the descriptor table); outputs are compact, in idx order
static int align_dp(mauve_ctx *c, const int64_t *idx, int64_t n, uint32_t *cols, int64_t *col_off, int64_t *score, int64_t *cells)
{
    AlignState &S = c->ast;
    const int N = S.N;
    c->dp_band_from = dp_band_from_of(&S.p);
    if (!idx) return dp_batch_run_desc(c, N, S.n_dp, S.desc.data(), &S.p.scoring, cols, col_off, score, cells);
    std::vector<DpSeqDesc> sub((size_t)n * N);
    for (int64_t k = 0; k < n; k++) {
        if (idx[k] < 0 || idx[k] >= S.n_dp) { c->err = "align_dp: interval index out of range"; return MAUVE_ERR_ARG; }
        std::copy(S.desc.begin() + idx[k] * N, S.desc.begin() + (idx[k] + 1) * N, sub.begin() + k * N);
    }
    return dp_batch_run_desc(c, N, n, sub.data(), &S.p.scoring, cols, col_off, score, cells);
}

// anchor_length / anchor_start / anchor_lcb of the result: depends on the chains only
static void fill_anchor_table(mauve_ctx *c)
{
    AlignState &S = c->ast; AlignResult &R = c->res;
    const int N = S.N;
    R.anchor_length.resize((size_t)S.n_anchor); R.anchor_start.resize((size_t)S.n_anchor * N); R.anchor_lcb.resize((size_t)S.n_anchor);
    size_t a = 0;
    for (int64_t l = 0; l < S.nl; l++) {
        const MatchVec &ch = S.chains[(size_t)l];
        for (size_t i = 0; i < ch.size(); i++, a++) {
            R.anchor_length[a] = ch.len(i); R.anchor_lcb[a] = l;
            std::copy(ch.st(i), ch.st(i) + N, &R.anchor_start[a * N]);
        }
    }
    S.anchor_table_done = true;
}

static int align_finish(mauve_ctx *c, const uint32_t *dcols, const int64_t *dcol_off, const int64_t *dscore, int64_t cells,
                        mauve_align_sizes *sizes)
{
    AlignState &S = c->ast;
    AlignResult &R = c->res;
    const int N = S.N; const int64_t nl = S.nl, n_dp = S.n_dp; const uint32_t full = S.full;
    const mauve_params *p = &S.p;
    std::vector<MatchVec> &chains = S.chains;
    const double t4 = now_ms();
    c->stage.dp_ms = t4 - S.t_dp0;
    // Host helpers (opt-in, MAUVE_HOST_THREADS > 1; workers.hpp) are only awake during the assembly.
    SpinPool::Armed helpers(c->pool);

    // ---- assemble the interval table ----
    // pass 1 (sequential, light): where every anchor and every stretch goes in the column array
    typedef AlignState::Item Item;
    std::vector<Item> &items = S.items; items.resize((size_t)S.n_anchor);
    R.col_off.clear();
    R.lcb_left.assign((size_t)nl * N, 0); R.lcb_right.assign((size_t)nl * N, 0);
    R.dp_score.assign((size_t)nl, 0);
    int64_t col = 0;
    {
        size_t gi = 0, ai = 0;
        for (int64_t l = 0; l < nl; l++) {
            const MatchVec &ch = chains[(size_t)l];
            R.col_off.push_back(col);
            for (size_t i = 0; i < ch.size(); i++) {
                Item it; it.lcb = l; it.idx = (uint32_t)i; it.col0 = col; it.gap = -1;
                col += ch.len(i);
                if (gi < S.gaps.size() && S.gaps[gi].lcb == l && S.gaps[gi].idx == (int64_t)i) {
                    const AlignState::GapRef &gr = S.gaps[gi];
                    it.gap = (int64_t)gi++;
                    if (gr.dp) { col += dcol_off[(size_t)gr.dp_slot + 1] - dcol_off[(size_t)gr.dp_slot]; R.dp_score[(size_t)l] += dscore[(size_t)gr.dp_slot]; }
                    else col += gr.tot;
                }
                items[ai++] = it;
            }
            // LCB extent: anchors are ordered, so the ends come from the first and last anchor
            if (ch.size()) {
                const size_t last = ch.size() - 1;
                for (int g = 0; g < N; g++) {
                    const int64_t s0 = ch.st(0)[g], s1 = ch.st(last)[g];
                    int64_t le, re;
                    if (s0 > 0) { le = s0; re = s1 + ch.len(last) - 1; }
                    else { le = -s1; re = -s0 + ch.len(0) - 1; }
                    R.lcb_left[(size_t)l * N + g] = s0 < 0 ? -le : le;
                    R.lcb_right[(size_t)l * N + g] = s0 < 0 ? -re : re;
                }
            }
        }
    }
    const double ta1 = now_ms();
    // The column array is a capacity buffer kept across calls in the "all anchors" state: every word outside the
    // ranges the previous call wrote gap columns into equals `full`.  Anchors are >80 % of the columns, so a call
    // restores the previous gap ranges and writes its own instead of filling tens of MB.
    const size_t need = (size_t)col + (size_t)(p->add_unaligned ? S.sum : 0);
    if (R.cols.size() < need || R.cols_fill != full) {
        if (R.cols.size() < need) R.cols.resize(need);
        std::fill(R.cols.begin(), R.cols.end(), full);
        R.cols_fill = full; R.cols_dirty.clear();
    } else {
        c->pool->parallel_for((int64_t)R.cols_dirty.size(), 1024, [&](int64_t b, int64_t e) {
            for (int64_t i = b; i < e; i++) {
                const auto &d = R.cols_dirty[(size_t)i];
                std::fill(R.cols.begin() + d.first, R.cols.begin() + d.first + d.second, full);
            }
        });
        R.cols_dirty.clear();
    }
    const double ta2 = now_ms();
    if (!S.anchor_table_done) fill_anchor_table(c);
    // pass 2: every anchor writes its record and the stretch that follows it (independent writes, run on the host
    // helpers).  Dirty ranges: one slot per anchor, length 0 where no gap follows.
    {
        uint32_t *out = R.cols.data();
        R.cols_dirty.assign((size_t)S.n_anchor, std::pair<size_t, size_t>(0, 0));
        c->pool->parallel_for(S.n_anchor, 1024, [&](int64_t ab, int64_t ae) {
            for (int64_t a = ab; a < ae; a++) {
                const Item &it = items[(size_t)a];
                const MatchVec &ch = chains[(size_t)it.lcb];
                const int64_t alen = ch.len(it.idx);
                uint32_t *o = out + it.col0 + alen;            // the anchor's own columns already hold `full`
                if (it.gap >= 0) {
                    const AlignState::GapRef &gr = S.gaps[(size_t)it.gap];
                    const size_t glen = (size_t)(gr.dp ? dcol_off[(size_t)gr.dp_slot + 1] - dcol_off[(size_t)gr.dp_slot] : gr.tot);
                    R.cols_dirty[(size_t)a] = {(size_t)(o - out), glen};
                    if (gr.dp) std::copy(dcols + dcol_off[(size_t)gr.dp_slot], dcols + dcol_off[(size_t)gr.dp_slot + 1], o);
                    else for (int g = 0; g < N; g++) {
                        int64_t lo, ln; bool rv;
                        gap_of(ch.rec(it.idx), ch.rec(it.idx + 1), g, lo, ln, rv);
                        std::fill(o, o + ln, 1u << g); o += ln;
                    }
                }
            }
        });
    }
    const double ta3 = now_ms();
    size_t ncols = (size_t)col;
    int64_t niv = nl;
    R.iv_left.assign((size_t)nl * N, 0); R.iv_right.assign((size_t)nl * N, 0); R.iv_reverse.assign((size_t)nl * N, 0);
    for (int64_t i = 0; i < nl * N; i++) {
        R.iv_left[(size_t)i] = std::llabs(R.lcb_left[(size_t)i]);
        R.iv_right[(size_t)i] = std::llabs(R.lcb_right[(size_t)i]);
        R.iv_reverse[(size_t)i] = R.lcb_left[(size_t)i] < 0;
    }
    if (p->add_unaligned) {
        for (int g = 0; g < N; g++) {
            std::vector<std::pair<int64_t, int64_t>> sp;
            for (int64_t l = 0; l < nl; l++) if (R.iv_left[(size_t)l * N + g]) sp.push_back({R.iv_left[(size_t)l * N + g], R.iv_right[(size_t)l * N + g]});
            std::sort(sp.begin(), sp.end());
            int64_t cur = 1;
            for (size_t i = 0; i <= sp.size(); i++) {
                int64_t lo = cur, hi = i < sp.size() ? sp[i].first - 1 : c->lens[g];
                if (hi >= lo) {
                    R.col_off.push_back((int64_t)ncols);
                    std::fill(R.cols.begin() + ncols, R.cols.begin() + ncols + (size_t)(hi - lo + 1), 1u << g);
                    R.cols_dirty.push_back({ncols, (size_t)(hi - lo + 1)});
                    ncols += (size_t)(hi - lo + 1);
                    for (int h = 0; h < N; h++) { R.iv_left.push_back(h == g ? lo : 0); R.iv_right.push_back(h == g ? hi : 0); R.iv_reverse.push_back(0); }
                    R.dp_score.push_back(0);
                    niv++;
                }
                if (i < sp.size() && sp[i].second + 1 > cur) cur = sp[i].second + 1;
            }
        }
    }
    R.col_off.push_back((int64_t)ncols);
    R.n_cols = ncols;
    R.sz.n_mums = S.nm; R.sz.n_lcb = nl; R.sz.n_anchor = S.n_anchor; R.sz.n_iv = niv; R.sz.n_cols = (int64_t)ncols;
    R.sz.n_gap_dp = n_dp; R.sz.n_dp_cells = cells;
    *sizes = R.sz;
    const double t5 = now_ms();
    if (g_trace_pipeline) fprintf(stderr, "[trace] assemble: layout %.3f ms, restore %.3f, anchors+gaps %.3f, unaligned+sizes %.3f\n", ta1 - t4, ta2 - ta1, ta3 - ta2, t5 - ta3);
    c->stage.assemble_ms = t5 - t4;
    c->stage.total_ms = t5 - S.t0;
    S.open = false;
    return MAUVE_OK;
}

// Aligner::align resumed from a list of LCBs (an IntervalList read back from an .mln file, mauveAligner.cpp:705-722; the matches
// of one LCB handed to Aligner::align again, :723-744 --realign-lcb): the chains are the caller's, no seed pass, no overlap
// elimination, no breakpoint elimination, no LCB extension; recursive anchoring and the gapped alignment of every inter-anchor
// interval run as in mauve_align.
static int align_begin_lcbs(mauve_ctx *c, const mauve_params *p, std::vector<MatchVec> &lcbs)
{
    const int N = c->nseq;
    AlignState &S = c->ast;
    S.reset();
    c->rec_flags.clear();                 // flags of an earlier call that ended before its recursion must not be taken for this one's
    S.p = *p; S.N = N; S.t0 = now_ms();
    AlignResult &R = c->res;
    R.sz = mauve_align_sizes();
    R.mum_length.clear(); R.mum_start.clear(); R.lcb_left.clear(); R.lcb_right.clear(); R.lcb_weight.clear();
    R.anchor_length.clear(); R.anchor_start.clear(); R.anchor_lcb.clear(); R.iv_left.clear(); R.iv_right.clear();
    R.iv_reverse.clear(); R.col_off.clear(); R.n_cols = 0; R.dp_score.clear();
    R.dev_pending = false; R.cols_pending = false; R.stale = false; R.genomes_replaced = false; R.dev_na = 0; R.dev_nm = 0; R.cols_ext = nullptr;
    memset(&c->stage, 0, sizeof c->stage);
    c->shadow = nullptr;
    int64_t sum = 0; for (int g = 0; g < N; g++) sum += c->lens[g];
    S.sum = sum;
    int w = p->seed_weight > 0 ? p->seed_weight : mauve_default_seed_weight(sum / N);
    if (p->seed_pattern) w = mauve_seed_weight(p->seed_pattern);
    S.full = N >= 32 ? 0xffffffffu : ((1u << N) - 1);
    S.nm = 0; S.nl = (int64_t)lcbs.size();
    S.chains.swap(lcbs);
    R.lcb_weight.assign((size_t)S.nl, 0);
    for (int64_t l = 0; l < S.nl; l++) for (size_t i = 0; i < S.chains[(size_t)l].size(); i++) R.lcb_weight[(size_t)l] += S.chains[(size_t)l].len(i) * N;
    const double t2 = now_ms();
    if (p->recursive) { const int rc = recursive_anchoring(c, p, w, S.chains, N, nullptr); if (rc) return rc; }
    c->stage.recurse_ms = now_ms() - t2;
    for (int64_t l = 0; l < S.nl; l++) {
        const MatchVec &ch = S.chains[(size_t)l];
        S.n_anchor += (int64_t)ch.size();
        for (size_t i = 0; i < ch.size(); i++) S.anchor_cols += ch.len(i);
    }
    S.t_dp0 = now_ms();
    S.open = true;
    return MAUVE_OK;
}

// the main pass's match list that was set aside in HBM (S.mums_kept) -> back in place and into the result's host tables
static int kept_mums_to_host(mauve_ctx *c)
{
    AlignState &S = c->ast; AlignResult &R = c->res;
    if (S.mums_kept < 0) return MAUVE_OK;
    std::swap(c->sorted_rec, c->sorted_rec_keep);
    const size_t n = (size_t)S.mums_kept, N = (size_t)S.N;
    S.mums_kept = -1;
    HIPCHK(c, c->pin_tab.ensure(n * (1 + N) * 8 + 64));
    if (n) HIPCHK(c, hipMemcpy(c->pin_tab.p, c->sorted_rec.p, n * (1 + N) * 8, hipMemcpyDeviceToHost));
    const int64_t *hm = c->pin_tab.as<int64_t>();
    R.mum_length.assign(hm, hm + n); R.mum_start.assign(hm + n, hm + n * (1 + N));
    return MAUVE_OK;
}

// The chains are on the host (S.chains: recursion, LCB extension, a small or tied list, a caller's LCBs): the anchors go up
// once, interval table, DP and -- unless MAUVE_HOST_TAIL -- the assembly run on the device.
static int align_tail_host_chains(mauve_ctx *c, mauve_align_sizes *sizes)
{
    AlignState &S = c->ast;
    static const bool no_shadow = getenv("MAUVE_NO_SHADOW") != nullptr;
    int64_t cells = 0;
    int rc;
    // the anchors in chain order as flat int32 records (page-locked), then everything up to the DP columns on the device
    const int N = S.N; const int64_t na = S.n_anchor;
    HIPCHK(c, c->pin_anch.ensure((size_t)na * (3 + (size_t)N) * 4 + 64));
    int32_t *h_len = c->pin_anch.as<int32_t>(), *h_st = h_len + na, *h_lcb = h_st + (size_t)na * N, *gapcode = h_lcb + na;
    {
        size_t a = 0;
        for (int64_t l = 0; l < S.nl; l++) {
            const MatchVec &ch = S.chains[(size_t)l];
            for (size_t i = 0; i < ch.size(); i++, a++) {
                const int64_t *r = ch.rec(i);
                h_len[a] = (int32_t)r[0]; h_lcb[a] = (int32_t)l;
                for (int g = 0; g < N; g++) h_st[a * N + g] = (int32_t)r[1 + g];
            }
        }
    }
    if (!no_shadow && na) c->shadow = [c]() { fill_anchor_table(c); };               // runs while the DP kernels do
    c->dp_band_from = dp_band_from_of(&S.p);
    static const bool host_tail = getenv("MAUVE_HOST_TAIL") != nullptr;
    if (!host_tail && na >= 2) {
        // The chains were on the host (recursion, LCB extension, a small or tied list), but nothing after them has to be:
        // the anchors go up, the DP results stay in HBM and the interval table is assembled there (assemble_dev.hip).
        rc = dp_run_from_anchors(c, N, na, h_len, h_st, h_lcb, S.p.gapped, dp_len_limit(&S.p), &S.p.scoring, nullptr, &S.n_dp, &S.code_total,
                                 nullptr, S.dcol_off, S.dscore, &cells, 2);
        if (c->shadow) { std::function<void()> f; f.swap(c->shadow); f(); }
        if (rc) return rc;
        return assemble_device(c, na, cells, sizes, true);
    }
    { const int rck = kept_mums_to_host(c); if (rck) return rck; }        // (the host assembly fills every table on the host)
    rc = dp_run_from_anchors(c, N, na, h_len, h_st, h_lcb, S.p.gapped, dp_len_limit(&S.p), &S.p.scoring, gapcode, &S.n_dp, &S.code_total,
                             &c->pin_dcols, S.dcol_off, S.dscore, &cells);
    c->shadow = nullptr;
    if (rc) return rc;
    {   // the gap list of the assembly, from the gap codes
        size_t a = 0;
        for (int64_t l = 0; l < S.nl; l++) {
            const MatchVec &ch = S.chains[(size_t)l];
            for (size_t i = 0; i < ch.size(); i++, a++) {
                const int32_t code = gapcode[a];
                if (code == -1) continue;
                AlignState::GapRef gr; gr.lcb = l; gr.idx = (int64_t)i; gr.dp = code >= 0; gr.dp_slot = code >= 0 ? code : -1; gr.tot = 0;
                if (!gr.dp)
                    for (int g = 0; g < N; g++) { int64_t lo, ln; bool rv; gap_of(ch.rec(i), ch.rec(i + 1), g, lo, ln, rv); gr.tot += ln; }
                S.gaps.push_back(gr);
            }
        }
    }
    S.dscore.push_back(0);
    return align_finish(c, c->pin_dcols.as<uint32_t>(), S.dcol_off.data(), S.dscore.data(), cells, sizes);
}

extern "C" {

int mauve_align(mauve_ctx *c, const mauve_params *p, mauve_align_sizes *sizes)
{
    if (!c || !p || !sizes) return MAUVE_ERR_ARG;
    if (c->nseq < 2) { c->err = "align: at least two genomes required"; return MAUVE_ERR_STATE; }
    HIPCHK(c, hipSetDevice(c->device));
    static const bool host_front = getenv("MAUVE_HOST_DP_FRONT") != nullptr;      // A/B switch
    static const bool no_shadow = getenv("MAUVE_NO_SHADOW") != nullptr;
    int rc = align_begin(c, p, !host_front, nullptr, !host_front);
    if (rc) return rc;
    AlignState &S = c->ast;
    int64_t cells = 0;
    if (S.dev_tail) {
        // chains, DP and assembly on the device: anchors from chain_order_device, results left in HBM until they are fetched
        const int32_t *d_len = S.dv_len, *d_st = S.dv_st, *d_lcb = S.dv_lcb;
        c->dp_band_from = dp_band_from_of(&S.p);
        rc = dp_run_from_anchors(c, S.N, S.n_anchor, d_len, d_st, d_lcb, S.p.gapped, dp_len_limit(&S.p), &S.p.scoring, nullptr, &S.n_dp, &S.code_total,
                                 nullptr, S.dcol_off, S.dscore, &cells, 1);
        if (rc) return rc;
        return assemble_device(c, S.n_anchor, cells, sizes);
    }
    if (host_front) {
        HIPCHK(c, c->pin_dcols.ensure(((size_t)S.code_total + 1) * sizeof(uint32_t)));
        uint32_t *dcols = c->pin_dcols.as<uint32_t>();
        S.dcol_off.assign((size_t)S.n_dp + 1, 0); S.dscore.assign((size_t)S.n_dp + 1, 0);
        if (!no_shadow && S.n_dp) c->shadow = [c]() { fill_anchor_table(c); };       // runs while the DP kernels do
        rc = align_dp(c, nullptr, S.n_dp, dcols, S.dcol_off.data(), S.dscore.data(), &cells);
        c->shadow = nullptr;
        if (rc) return rc;
        return align_finish(c, dcols, S.dcol_off.data(), S.dscore.data(), cells, sizes);
    }
    return align_tail_host_chains(c, sizes);
}

int mauve_align_lcbs(mauve_ctx *c, const mauve_params *p, int64_t n, const int64_t *length, const int64_t *start, const int64_t *lcb, mauve_align_sizes *sizes)
{
    if (!c || !p || !sizes) return MAUVE_ERR_ARG;
    if (c->nseq < 2) { c->err = "align_lcbs: at least two genomes required"; return MAUVE_ERR_STATE; }
    if (n < 0 || (n && (!length || !start || !lcb))) { c->err = "align_lcbs: bad anchor list"; return MAUVE_ERR_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    { int rcm = materialize_tables(c); if (rcm) return rcm; }
    const int N = c->nseq;
    int64_t nl = 0;
    for (int64_t i = 0; i < n; i++) {
        if (lcb[i] < 0 || lcb[i] >= n) { c->err = "align_lcbs: LCB ids must be 0 .. n_lcb-1"; return MAUVE_ERR_ARG; }
        nl = std::max(nl, lcb[i] + 1);
        if (length[i] <= 0) { c->err = "align_lcbs: anchor of length <= 0"; return MAUVE_ERR_ARG; }
        for (int g = 0; g < N; g++) {
            const int64_t s = start[i * N + g];
            if (!s) { c->err = "align_lcbs: anchors have a component in every genome"; return MAUVE_ERR_ARG; }
            if (std::llabs(s) + length[i] - 1 > c->lens[(size_t)g]) { c->err = "align_lcbs: anchor outside its genome"; return MAUVE_ERR_ARG; }
        }
        if (start[i * N] < 0) { c->err = "align_lcbs: anchors are forward in genome 0 (Match::Invert them first)"; return MAUVE_ERR_ARG; }
    }
    std::vector<MatchVec> chains((size_t)nl, MatchVec(N));
    for (int64_t i = 0; i < n; i++) chains[(size_t)lcb[i]].push(length[i], start + i * N);
    for (int64_t l = 0; l < nl; l++) {
        MatchVec &ch = chains[(size_t)l];
        if (ch.empty()) { c->err = "align_lcbs: an LCB id without anchors"; return MAUVE_ERR_ARG; }
        ch.sort_by_start0();
        // one collinear chain: the same strand relation all along, and in every genome the anchors follow each other without overlap
        for (size_t i = 0; i + 1 < ch.size(); i++)
            for (int g = 0; g < N; g++) {
                const int64_t a = ch.st(i)[g], b = ch.st(i + 1)[g];
                const bool ok = (a > 0) == (b > 0) && (a > 0 ? a + ch.len(i) <= b : -b + ch.len(i + 1) <= -a);
                if (!ok) { c->err = "align_lcbs: the anchors of an LCB must be collinear and free of overlaps"; return MAUVE_ERR_ARG; }
            }
    }
    int rc = align_begin_lcbs(c, p, chains);
    if (rc) return rc;
    return align_tail_host_chains(c, sizes);
}

// Aligner::align(MatchList&, ...) with the caller's own match list: chaining, recursion and gapped alignment as in
// mauve_align, no seed pass of its own
static int given_matches(mauve_ctx *c, int64_t n, const int64_t *length, const int64_t *start, MatchVec &mv)
{
    if (n < 0 || (n && (!length || !start))) { c->err = "align_matches: bad match list"; return MAUVE_ERR_ARG; }
    mv.N = c->nseq; mv.resize((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        if (length[i] <= 0) { c->err = "align_matches: match of length <= 0"; return MAUVE_ERR_ARG; }
        mv.len((size_t)i) = length[i];
        for (int g = 0; g < c->nseq; g++) {
            const int64_t s = start[i * c->nseq + g];
            if (s && std::llabs(s) + length[i] - 1 > c->lens[(size_t)g]) { c->err = "align_matches: match outside its genome"; return MAUVE_ERR_ARG; }
            mv.st((size_t)i)[g] = s;
        }
    }
    return MAUVE_OK;
}

int mauve_match_sp_scores(mauve_ctx *c, int64_t n, const int64_t *length, const int64_t *start, const mauve_scoring *sc, int64_t *scores)
{
    if (!c || !sc || n < 0 || (n && (!length || !start || !scores))) return MAUVE_ERR_ARG;
    if (c->nseq < 2) { c->err = "match_sp_scores: at least two genomes required"; return MAUVE_ERR_STATE; }
    HIPCHK(c, hipSetDevice(c->device));
    const int N = c->nseq;
    MatchVec m(N); m.resize((size_t)n);
    for (int64_t i = 0; i < n; i++) { m.len((size_t)i) = length[i]; std::copy(start + i * N, start + (i + 1) * N, m.st((size_t)i)); }
    std::vector<int64_t> out;
    int rc = match_sp_scores(c, m, nullptr, sc, out);
    if (rc) return rc;
    std::copy(out.begin(), out.end(), scores);
    return MAUVE_OK;
}

int mauve_align_matches(mauve_ctx *c, const mauve_params *p, int64_t n, const int64_t *length, const int64_t *start, mauve_align_sizes *sizes)
{
    if (!c || !p || !sizes) return MAUVE_ERR_ARG;
    if (c->nseq < 2) { c->err = "align: at least two genomes required"; return MAUVE_ERR_STATE; }
    HIPCHK(c, hipSetDevice(c->device));
    MatchVec mv;
    int rc = given_matches(c, n, length, start, mv);
    if (rc) return rc;
    rc = align_begin(c, p, false, &mv);
    if (rc) return rc;
    AlignState &S = c->ast;
    HIPCHK(c, c->pin_dcols.ensure(((size_t)S.code_total + 1) * sizeof(uint32_t)));
    uint32_t *dcols = c->pin_dcols.as<uint32_t>();
    S.dcol_off.assign((size_t)S.n_dp + 1, 0); S.dscore.assign((size_t)S.n_dp + 1, 0);
    int64_t cells = 0;
    rc = align_dp(c, nullptr, S.n_dp, dcols, S.dcol_off.data(), S.dscore.data(), &cells);
    if (rc) return rc;
    return align_finish(c, dcols, S.dcol_off.data(), S.dscore.data(), cells, sizes);
}

int mauve_align_begin_matches(mauve_ctx *c, const mauve_params *p, int64_t n, const int64_t *length, const int64_t *start, int64_t *n_dp,
                              int64_t *n_codes)
{
    if (!c || !p || !n_dp) return MAUVE_ERR_ARG;
    if (c->nseq < 2) { c->err = "align: at least two genomes required"; return MAUVE_ERR_STATE; }
    HIPCHK(c, hipSetDevice(c->device));
    MatchVec mv;
    int rc = given_matches(c, n, length, start, mv);
    if (rc) return rc;
    rc = align_begin(c, p, false, &mv);
    if (rc) return rc;
    *n_dp = c->ast.n_dp;
    if (n_codes) *n_codes = c->ast.code_total;
    return MAUVE_OK;
}

int mauve_align_dp_anchors(mauve_ctx *c, int64_t *left, int64_t *right)
{
    if (!c || !c->ast.open) { if (c) c->err = "align_dp_anchors: no alignment in progress"; return c ? MAUVE_ERR_STATE : MAUVE_ERR_ARG; }
    const AlignState &S = c->ast;
    const int N = S.N;
    for (const AlignState::GapRef &gr : S.gaps) {
        if (!gr.dp) continue;
        const MatchVec &ch = S.chains[(size_t)gr.lcb];
        if (left) std::copy(ch.rec((size_t)gr.idx), ch.rec((size_t)gr.idx) + 1 + N, left + gr.dp_slot * (1 + N));
        if (right) std::copy(ch.rec((size_t)gr.idx + 1), ch.rec((size_t)gr.idx + 1) + 1 + N, right + gr.dp_slot * (1 + N));
    }
    return MAUVE_OK;
}

// ---- sharded form of mauve_align: see the phase comment above -------------------------------------------------
int mauve_align_begin(mauve_ctx *c, const mauve_params *p, int64_t *n_dp, int64_t *n_codes)
{
    if (!c || !p || !n_dp) return MAUVE_ERR_ARG;
    if (c->nseq < 2) { c->err = "align: at least two genomes required"; return MAUVE_ERR_STATE; }
    HIPCHK(c, hipSetDevice(c->device));
    int rc = align_begin(c, p);
    if (rc) return rc;
    *n_dp = c->ast.n_dp;
    if (n_codes) *n_codes = c->ast.code_total;
    return MAUVE_OK;
}

int mauve_align_dp_cost(mauve_ctx *c, int64_t *cost, int64_t *max_cols)
{
    if (!c || !c->ast.open) { if (c) c->err = "align_dp_cost: no alignment in progress"; return c ? MAUVE_ERR_STATE : MAUVE_ERR_ARG; }
    const AlignState &S = c->ast;
    for (int64_t k = 0; k < S.n_dp; k++) {
        int64_t m = 0, cells = 0, tot = 0;
        for (int g = 0; g < S.N; g++) {
            const int64_t n = S.desc[(size_t)(k * S.N + g)].len;
            tot += n;
            if (!n) continue;
            if (!m) { m = n; continue; }
            cells += m * n; m += n;
        }
        if (cost) cost[k] = cells;
        if (max_cols) max_cols[k] = tot;
    }
    return MAUVE_OK;
}

int mauve_align_dp(mauve_ctx *c, const int64_t *idx, int64_t n, uint32_t *cols, int64_t *col_off, int64_t *score, int64_t *cells)
{
    if (!c || !col_off || (n && (!idx || !cols))) return MAUVE_ERR_ARG;
    if (!c->ast.open) { c->err = "align_dp: no alignment in progress"; return MAUVE_ERR_STATE; }
    HIPCHK(c, hipSetDevice(c->device));
    return align_dp(c, idx, n, cols, col_off, score, cells);
}

int mauve_align_finish(mauve_ctx *c, const uint32_t *cols, const int64_t *col_off, const int64_t *score, int64_t cells,
                       mauve_align_sizes *sizes)
{
    if (!c || !sizes || !col_off) return MAUVE_ERR_ARG;
    if (!c->ast.open) { c->err = "align_finish: no alignment in progress"; return MAUVE_ERR_STATE; }
    return align_finish(c, cols, col_off, score, cells, sizes);
}

#define CPY(dst, vec) do { if ((dst) && !(vec).empty()) memcpy((dst), (vec).data(), (vec).size() * sizeof((vec)[0])); } while (0)

int mauve_align_fetch(mauve_ctx *c, int64_t *mum_length, int64_t *mum_start, int64_t *lcb_left, int64_t *lcb_right,
                      int64_t *lcb_weight, int64_t *anchor_length, int64_t *anchor_start, int64_t *anchor_lcb,
                      int64_t *iv_left, int64_t *iv_right, int8_t *iv_reverse, int64_t *col_off, uint32_t *cols,
                      int64_t *dp_score)
{
    if (!c) return MAUVE_ERR_ARG;
    if (c->res.stale) { c->err = "align_fetch: the genomes were replaced after this alignment was made; its device-resident part is gone"; return MAUVE_ERR_STATE; }
    // the two bulk tables of a device-assembled result: straight from HBM into page-locked caller buffers when all of them are
    // asked for and are such; through the context's host copy otherwise
    bool direct = false;
    if (c->res.dev_pending && mum_length && mum_start && anchor_length && anchor_start && anchor_lcb) {
        int rcd = MAUVE_OK;
        direct = fetch_tables_direct(c, mum_length, mum_start, anchor_length, anchor_start, anchor_lcb, &rcd);
        if (rcd) return rcd;
    }
    if (!direct) { int rcm = materialize_tables(c); if (rcm) return rcm; }
    const AlignResult &R = c->res;
    if (!direct || R.dev_nm == 0) { CPY(mum_length, R.mum_length); CPY(mum_start, R.mum_start); }        // (direct: only what was still on the device)
    if (!direct || R.dev_na == 0) { CPY(anchor_length, R.anchor_length); CPY(anchor_start, R.anchor_start); CPY(anchor_lcb, R.anchor_lcb); }
    CPY(lcb_left, R.lcb_left); CPY(lcb_right, R.lcb_right); CPY(lcb_weight, R.lcb_weight);
    CPY(iv_left, R.iv_left); CPY(iv_right, R.iv_right); CPY(iv_reverse, R.iv_reverse);
    CPY(col_off, R.col_off); CPY(dp_score, R.dp_score);
    if (cols && R.n_cols) { const int rcc = fetch_columns(c, cols); if (rcc) return rcc; }
    if (direct) HIPCHK(c, hipStreamSynchronize(c->stream));         // the table copies are in flight on the same stream
    return MAUVE_OK;
}

// The same result in the narrowest types that hold it (mauve_hip.h): one byte per column for up to 8 genomes (two up to 16), 32-bit match and
// anchor tables (a context holds fewer than 2^31 bases).  The copy out is what the host-to-host metric pays for beyond the pass itself: at
// C5 the uint32 columns alone are 490 MB for a two-genome alignment.
int mauve_align_fetch_compact(mauve_ctx *c, int col_bytes, int32_t *mum_length, int32_t *mum_start, int64_t *lcb_left, int64_t *lcb_right,
                              int64_t *lcb_weight, int32_t *anchor_length, int32_t *anchor_start, int32_t *anchor_lcb,
                              int64_t *iv_left, int64_t *iv_right, int8_t *iv_reverse, int64_t *col_off, void *cols, int64_t *dp_score)
{
    if (!c) return MAUVE_ERR_ARG;
    if (col_bytes != 1 && col_bytes != 2 && col_bytes != 4) { c->err = "align_fetch_compact: col_bytes must be 1, 2 or 4"; return MAUVE_ERR_ARG; }
    if (c->res.stale) { c->err = "align_fetch_compact: the genomes were replaced after this alignment was made; its device-resident part is gone"; return MAUVE_ERR_STATE; }
    const int64_t n_iv = c->res.sz.n_iv;
    const int Nres = n_iv ? (int)(c->res.iv_left.size() / (size_t)n_iv) : c->nseq;
    if (Nres > 8 * col_bytes) { c->err = "align_fetch_compact: a column needs one bit per genome (nseq <= 8 x col_bytes)"; return MAUVE_ERR_ARG; }
    bool tables_done = false, cols_done = false; int rcd = MAUVE_OK;
    (void)fetch_compact_direct(c, col_bytes, mum_length, mum_start, anchor_length, anchor_start, anchor_lcb, cols, &tables_done, &cols_done, &rcd);
    if (rcd) return rcd;
    if (!tables_done) { int rcm = materialize_tables(c); if (rcm) return rcm; }
    const AlignResult &R = c->res;
    auto narrow = [](int32_t *dst, const std::vector<int64_t> &v) { if (dst) for (size_t i = 0; i < v.size(); i++) dst[i] = (int32_t)v[i]; };
    if (!tables_done || R.dev_nm == 0) { narrow(mum_length, R.mum_length); narrow(mum_start, R.mum_start); }
    if (!tables_done || R.dev_na == 0) { narrow(anchor_length, R.anchor_length); narrow(anchor_start, R.anchor_start); narrow(anchor_lcb, R.anchor_lcb); }
    CPY(lcb_left, R.lcb_left); CPY(lcb_right, R.lcb_right); CPY(lcb_weight, R.lcb_weight);
    CPY(iv_left, R.iv_left); CPY(iv_right, R.iv_right); CPY(iv_reverse, R.iv_reverse);
    CPY(col_off, R.col_off); CPY(dp_score, R.dp_score);
    if (cols && R.n_cols && !cols_done) {
        { int rcm = materialize_result(c); if (rcm) return rcm; }
        const uint32_t *src = c->res.cols_data(); const size_t n = c->res.n_cols;
        if (col_bytes == 4) memcpy(cols, src, n * 4);
        else if (col_bytes == 2) { uint16_t *d = static_cast<uint16_t *>(cols); for (size_t i = 0; i < n; i++) d[i] = (uint16_t)src[i]; }
        else { uint8_t *d = static_cast<uint8_t *>(cols); for (size_t i = 0; i < n; i++) d[i] = (uint8_t)src[i]; }
    }
    return MAUVE_OK;
}

// IntervalList::WriteStandardAlignment [EXT] (mauveAligner.cpp:746-760); text format pinned by the in-tree
// writer mfa2xmfa.cpp:64 (header), :89-91 (#Sequence lines), :104-115 (entry line, 80-column rows, '=').
int mauve_write_xmfa(mauve_ctx *c, const char *const *names, char *buf, int64_t *len)
{
    if (!c || !len) return MAUVE_ERR_ARG;
    if (c->res.stale || c->res.genomes_replaced) { c->err = "write_xmfa: the genomes were replaced after this alignment was made"; return MAUVE_ERR_STATE; }
    { int rcm = materialize_result(c); if (rcm) return rcm; }
    { int rcm = host_genomes(c); if (rcm) return rcm; }
    const AlignResult &R = c->res;
    const int N = c->nseq;
    const uint32_t *rcols = R.cols_data();
    static const char B[4] = {'A', 'C', 'G', 'T'};
    std::string out;
    out.reserve(R.n_cols * (size_t)N / 2 + 4096);
    char line[600];
    out += "#FormatVersion Mauve1\n";
    for (int g = 0; g < N; g++) {
        snprintf(line, sizeof line, "#Sequence%dFile\t%s\n#Sequence%dEntry\t%d\n#Sequence%dFormat\tFastA\n", g + 1,
                 names && names[g] ? names[g] : "", g + 1, g + 1, g + 1);
        out += line;
    }
    std::string row;
    for (int64_t iv = 0; iv < R.sz.n_iv; iv++) {
        const int64_t c0 = R.col_off[(size_t)iv], nc = R.col_off[(size_t)iv + 1] - c0;
        for (int g = 0; g < N; g++) {
            const int64_t le = R.iv_left[(size_t)iv * N + g], re = R.iv_right[(size_t)iv * N + g];
            if (!le) continue;
            const bool rev = R.iv_reverse[(size_t)iv * N + g] != 0;
            int64_t nxt = rev ? re : le;
            row.resize((size_t)nc);
            const uint64_t *w = c->host_packed[(size_t)g];
            for (int64_t k = 0; k < nc; k++) {
                if (rcols[(size_t)(c0 + k)] >> g & 1) {
                    uint8_t b = base_at(w, nxt - 1);
                    row[(size_t)k] = rev ? B[3 - b] : B[b];
                    if (c->has_invalid && (c->h_invalid[c->base_mask_off[(size_t)g] + (size_t)((nxt - 1) >> 6)] >> ((nxt - 1) & 63) & 1)) row[(size_t)k] = 'N';
                    nxt += rev ? -1 : 1;
                } else row[(size_t)k] = '-';
            }
            snprintf(line, sizeof line, "> %d:%lld-%lld %c %s\n", g + 1, (long long)le, (long long)re, rev ? '-' : '+',
                     names && names[g] ? names[g] : "");
            out += line;
            for (int64_t pos = 0; pos < nc; pos += 80) { out.append(row, (size_t)pos, (size_t)std::min<int64_t>(80, nc - pos)); out += '\n'; }
        }
        out += "=\n";
    }
    if (!buf) { *len = (int64_t)out.size() + 1; return MAUVE_OK; }
    if (*len < (int64_t)out.size() + 1) { c->err = "write_xmfa: buffer too small"; return MAUVE_ERR_ARG; }
    memcpy(buf, out.c_str(), out.size() + 1);
    *len = (int64_t)out.size() + 1;
    return MAUVE_OK;
}

}  // extern "C"
