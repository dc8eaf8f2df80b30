// progressive.cpp -- guide tree and guide-tree recursive anchoring behind mauve_progressive_align():
// the stand-in for ProgressiveAligner::align(seq_table, interval_list) [EXT] (progressiveMauve.cpp:575-710) and
// for the distance matrix / guide tree of mauveAligner.cpp:616-623.  Frozen spec: DESIGN.md S9.
//
//   1. pairwise matches of every genome pair (PairwiseMatchFinder rule, one sorted mer list on the device),
//      similarity = matched bases / shorter genome, integer distance in ppm, UPGMA with integer averaging;
//   2. the root of the tree aligns what all genomes share (the Aligner::align path); every internal node below
//      aligns, among its own genomes only, the bases no ancestor has placed.  Placed bases are a 1-bit-per-base
//      mask on the device: windows touching them are invalid for seeding and extension (seed_pass.hip), chains
//      are cut where the stretch between two anchors touches a placed base or is too long for the gapped
//      aligner, so those bases flow down to the subtree that shares them;
//   3. what is left at the leaves becomes single-genome intervals.
// Every node is one pass of the same device pipeline (seed pass, batched recursive anchoring, batched DP).
#include "common.hpp"
#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstring>
#include <map>

int recursive_anchoring(mauve_ctx *c, const mauve_params *p, int w0, std::vector<MatchVec> &chains, int N, const int *gmap);

namespace {

// The bases of one genome that no block has taken yet: an ordered map start -> end of free stretches (1-based,
// inclusive, disjoint).  Blocks are carved out one by one; a position is identified by the start of the stretch it lies
// in (0 = placed already).
struct FreePool {
    std::map<int64_t, int64_t> free_;
    void reset(int64_t len) { free_.clear(); if (len >= 1) free_[1] = len; }
    bool whole(int64_t len) const { return free_.size() == 1 && free_.begin()->first == 1 && free_.begin()->second == len; }
    int64_t bases() const { int64_t t = 0; for (const auto &kv : free_) t += kv.second - kv.first + 1; return t; }
    int64_t stretch_of(int64_t pos) const
    {
        auto it = free_.upper_bound(pos);
        if (it == free_.begin()) return 0;
        --it;
        return pos <= it->second ? it->first : 0;
    }
    void carve(int64_t l, int64_t h)              // free minus [l, h]
    {
        if (h < l) return;
        auto it = free_.upper_bound(l);
        if (it != free_.begin()) --it;
        while (it != free_.end() && it->first <= h) {
            const int64_t a = it->first, e = it->second;
            if (e < l) { ++it; continue; }
            it = free_.erase(it);
            if (a < l) free_[a] = l - 1;
            if (e > h) { free_[h + 1] = e; break; }
        }
    }
};

inline void gap_of(const int64_t *a, const int64_t *b, int g, int64_t &lo, int64_t &len, bool &rev)
{
    const int64_t sa = a[1 + g], sb = b[1 + g];
    int64_t hi;
    if (sa > 0) { lo = sa + a[0]; hi = sb - 1; rev = false; }
    else { lo = -sb + b[0]; hi = -sa - 1; rev = true; }
    len = hi - lo + 1; if (len < 0) len = 0;
}

struct Prog {
    mauve_ctx *c; const mauve_params *p; int N;
    std::vector<int32_t> left, right;
    std::vector<int64_t> dist;         // [N*N] ppm, filled when the node weights are scaled
    std::vector<int64_t> bpd;          // [N*N] ppm breakpoint distance, filled when bp_dist_scale_ppm is set as well
    std::vector<FreePool> rest;        // per genome: bases not placed in any block yet
    AlignResult *R;
    int64_t n_gap_dp = 0, n_cells = 0, n_anchor = 0, n_multi = 0;
    double t_seed = 0, t_chain = 0, t_rec = 0, t_dp = 0, t_blocks = 0;      // stage times summed over the nodes (mauve_last_stage_times)
};

int leaves_of(const Prog &P, int node, std::vector<int> &out)
{
    if (P.left[node] < 0) { out.push_back(node); return 1; }
    return leaves_of(P, P.left[node], out) + leaves_of(P, P.right[node], out);
}

// upload the placed-base bitmap of the node's genomes (bit set = placed)
int upload_mask(Prog &P, const std::vector<int> &gm, GenomeSet &gs)
{
    mauve_ctx *c = P.c;
    gs.mask_off.assign(gm.size(), 0);
    size_t words = 0;
    for (size_t j = 0; j < gm.size(); j++) { gs.mask_off[j] = words; words += (size_t)((c->lens[gm[j]] + 63) / 64) + 2; }
    std::vector<uint64_t> bits(words, ~0ULL);
    for (size_t j = 0; j < gm.size(); j++) {
        uint64_t *M = bits.data() + gs.mask_off[j];
        for (const auto &fr : P.rest[gm[j]].free_)
            for (int64_t b = fr.first - 1; b < fr.second;) {        // clear [lo-1, hi) word-wise
                const int64_t w = b >> 6, e = std::min<int64_t>(fr.second, (w + 1) << 6);
                const int n = (int)(e - b), sh = (int)(b & 63);
                const uint64_t m = (n == 64 ? ~0ULL : ((1ULL << n) - 1ULL)) << sh;
                M[w] &= ~m; b = e;
            }
    }
    // ambiguous bases stay unusable at every node; contig joins likewise (mauve_set_genomes_contigs)
    if (c->has_invalid)
        for (size_t j = 0; j < gm.size(); j++) {
            const size_t w = (size_t)((c->lens[gm[j]] + 63) / 64);
            for (size_t k = 0; k < w; k++) bits[gs.mask_off[j] + k] |= c->h_invalid[c->base_mask_off[(size_t)gm[j]] + k];
        }
    HIPCHK(c, c->placed_mask.ensure(words * 8));
    HIPCHK(c, hipMemcpyAsync(c->placed_mask.p, bits.data(), words * 8, hipMemcpyHostToDevice, c->stream));
    if (c->has_contigs) {
        std::vector<uint64_t> cm(words, 0);
        for (size_t j = 0; j < gm.size(); j++) {
            const size_t w = (size_t)((c->lens[gm[j]] + 63) / 64);
            for (size_t k = 0; k < w; k++) cm[gs.mask_off[j] + k] = c->h_contig[c->base_mask_off[(size_t)gm[j]] + k];
        }
        HIPCHK(c, c->node_cmask.ensure(words * 8));
        HIPCHK(c, hipMemcpy(c->node_cmask.p, cm.data(), words * 8, hipMemcpyHostToDevice));
        gs.cmask = &c->node_cmask;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    gs.vmask = &c->placed_mask;
    return MAUVE_OK;
}

int prog_node(Prog &P, int node)
{
    if (P.left[node] < 0) return MAUVE_OK;
    mauve_ctx *c = P.c; const mauve_params *p = P.p;
    std::vector<int> gm; leaves_of(P, node, gm);
    std::sort(gm.begin(), gm.end());
    const int n = (int)gm.size();
    AlignResult &R = *P.R;

    int64_t rest_len = 0;
    for (int j = 0; j < n; j++) rest_len += P.rest[gm[j]].bases();
    int w = p->seed_weight > 0 ? p->seed_weight : mauve_default_seed_weight(rest_len / n);
    uint64_t pat = p->seed_pattern ? p->seed_pattern : mauve_get_seed(w, p->seed_rank);
    if (!pat) { c->err = "progressive_align: no seed pattern for this weight/rank"; return MAUVE_ERR_ARG; }
    w = mauve_seed_weight(pat);
    const uint32_t full = n >= 32 ? 0xffffffffu : ((1u << n) - 1);

    GenomeSet gs; gs.buf = &c->genomes; gs.nseq = n;
    for (int j = 0; j < n; j++) { gs.lens.push_back(c->lens[gm[j]]); gs.word_off.push_back(c->word_off[gm[j]]); }
    bool any_placed = false;
    for (int j = 0; j < n; j++) if (!P.rest[gm[j]].whole(c->lens[gm[j]])) any_placed = true;
    if (any_placed) { int rc = upload_mask(P, gm, gs); if (rc) return rc; }

    static const bool trace = getenv("MAUVE_TRACE") != nullptr;
    const double tn0 = now_ms();
    int64_t nm = 0;
    MatchVec m(n);
    int rc;
    if (p->seed_family) {                                    // the node searches with the whole seed family
        rc = seed_family_matches(c, gs, w, p->mode, full, m);
        if (rc) return rc;
        nm = (int64_t)m.size();
    } else {
        rc = seedpass_run(c, gs, pat, p->mode, full, 1, nullptr, 0, &nm);
        if (rc) return rc;
        m.resize((size_t)nm);
        for (int64_t i = 0; i < nm; i++) {
            m.len((size_t)i) = c->match_len[(size_t)i];
            std::copy(&c->match_start[(size_t)i * n], &c->match_start[(size_t)i * n] + n, m.st((size_t)i));
        }
    }
    const double tn1 = now_ms();
    ChainOrders orders;
    int64_t lcbw = p->lcb_weight >= 0 ? p->lcb_weight * n / P.N : (int64_t)3 * w * n;
    // the node's minimum weight shrinks with the conservation distance between its two subtrees
    int64_t factor_ppm = 1000000;
    const bool scaled = p->weight_scaling && !P.dist.empty();
    if (scaled) {
        std::vector<int> la, lb;
        leaves_of(P, P.left[node], la); leaves_of(P, P.right[node], lb);
        int64_t sum = 0;
        for (int a : la) for (int b : lb) sum += P.dist[(size_t)a * P.N + b];
        const int64_t c_ppm = sum / ((int64_t)la.size() * (int64_t)lb.size());
        factor_ppm = std::max<int64_t>(0, 1000000 - (int64_t)p->conservation_scale_ppm * c_ppm / 1000000);
        if (!P.bpd.empty()) {                                 // the breakpoint-distance factor multiplies in
            int64_t bsum = 0;
            for (int a : la) for (int b : lb) bsum += P.bpd[(size_t)a * P.N + b];
            const int64_t f2 = std::max<int64_t>(0, 1000000 - (int64_t)p->bp_dist_scale_ppm * (bsum / ((int64_t)la.size() * (int64_t)lb.size())) / 1000000);
            factor_ppm = factor_ppm * f2 / 1000000;
        }
        lcbw = std::max(lcbw * factor_ppm / 1000000, p->min_scaled_penalty);
    }
    std::vector<int64_t> match_lcb; int64_t nl = 0;
    // The root's list (all genomes, tens of thousands of matches) is still in HBM in canonical order: overlap elimination and LCBs on the
    // device (chain_dev.hip), cropped records and labels back -- what mauve_align does.  Nodes below the root have short lists (and a
    // genome subset: the device chain takes its position width from the context's genomes), score-weighted LCBs need the matches' scores,
    // a list the device chain declines comes back as MAUVE_ERR_LIMIT: the host chain for those.  MAUVE_HOST_CHAIN: A/B switch.
    static const bool host_chain_env = getenv("MAUVE_HOST_CHAIN") != nullptr;
    bool chained = false;
    if (!host_chain_env && !p->seed_family && n == P.N && nm > 0 && c->dev_rec_n == nm && (p->lcb_scoring == MAUVE_LCB_SCORE_LENGTH || n <= 16)) {
        if (p->lcb_scoring == MAUVE_LCB_SCORE_SP) {          // the matches' scores are summed on the device too (ch_sp_scores, on the cropped records)
            int64_t minw = p->lcb_weight >= 0 ? p->lcb_weight * ((int64_t)n * (n - 1) / 2) / ((int64_t)P.N * (P.N - 1) / 2)
                                              : sp_default_min_weight(w, n, &p->scoring);
            if (scaled) minw = std::max(minw * factor_ppm / 1000000, p->min_scaled_penalty);
            rc = chain_device(c, n, minw, p->collinear != 0, m, match_lcb, nl, &p->scoring);
        } else
            rc = chain_device(c, n, lcbw, p->collinear != 0, m, match_lcb, nl);
        if (rc == MAUVE_OK) chained = true;
        else if (rc != MAUVE_ERR_LIMIT) return rc;
    }
    if (!chained) {
        host_eliminate_overlaps(m, &orders);
        if (p->lcb_scoring == MAUVE_LCB_SCORE_SP) {
            std::vector<int64_t> mw;
            rc = match_sp_scores(c, m, gm.data(), &p->scoring, mw);
            if (rc) return rc;
            // a given score threshold is for all N genomes: scaled by the node's share of the pairs
            int64_t minw = p->lcb_weight >= 0 ? p->lcb_weight * ((int64_t)n * (n - 1) / 2) / ((int64_t)P.N * (P.N - 1) / 2)
                                              : sp_default_min_weight(w, n, &p->scoring);
            if (scaled) minw = std::max(minw * factor_ppm / 1000000, p->min_scaled_penalty);
            host_lcb_chain(m, minw, p->collinear != 0, match_lcb, nl, &orders, mw.data());
        } else
            host_lcb_chain(m, lcbw, p->collinear != 0, match_lcb, nl, &orders);
    }
    if (trace) {
        int64_t surv = 0; for (size_t i = 0; i < m.size(); i++) if (match_lcb[i] >= 0) surv++;
        fprintf(stderr, "[trace] node %d (n=%d, w=%d): %lld n-way matches, %zu after overlap elimination, %lld lcbs, %lld anchors\n", node, n, w,
                (long long)nm, m.size(), (long long)nl, (long long)surv);
        if (getenv("MAUVE_TRACE_MATCHES"))
            for (size_t i = 0; i < m.size(); i++) { fprintf(stderr, "[trace]   m %zu len %lld lcb %lld:", i, (long long)m.len(i), (long long)match_lcb[i]); for (int j = 0; j < n; j++) fprintf(stderr, " %lld", (long long)m.st(i)[j]); fprintf(stderr, "\n"); }
    }
    // chains, cut wherever the stretch between two consecutive anchors touches an already placed base
    std::vector<MatchVec> pieces; int64_t cur_lcb = -1;
    for (size_t i = 0; i < m.size(); i++) {
        const int64_t l = match_lcb[i]; if (l < 0) continue;
        bool newp = l != cur_lcb;
        if (!newp) {
            const MatchVec &last = pieces.back();
            const int64_t *a = last.st(last.size() - 1);
            for (int j = 0; j < n && !newp; j++)
                if (P.rest[gm[j]].stretch_of(std::llabs(a[j])) != P.rest[gm[j]].stretch_of(std::llabs(m.st(i)[j]))) newp = true;
        }
        if (newp) { pieces.emplace_back(n); cur_lcb = l; }
        pieces.back().push(m.rec(i));
    }
    const double tn2 = now_ms();
    if (p->recursive) { rc = recursive_anchoring(c, p, w, pieces, n, gm.data()); if (rc) return rc; }
    const double tn3 = now_ms();
    // a stretch shared by >= 2 genomes that is too long for the gapped aligner, or that only a proper subset of
    // this node's genomes has (and that is long enough to be anchored), ends the block: its bases stay in the
    // pool for the subtree that shares them
    if (p->gapped) {
        std::vector<MatchVec> np2;
        for (const MatchVec &pc : pieces)
            for (size_t i = 0; i < pc.size(); i++) {
                bool split = i == 0;
                if (!split) {
                    int64_t mx = 0; int nonempty = 0, big = 0;
                    for (int j = 0; j < n; j++) { int64_t lo, ln; bool rv; gap_of(pc.rec(i - 1), pc.rec(i), j, lo, ln, rv); mx = std::max(mx, ln); nonempty += ln > 0; big += ln > p->min_recursive_gap; }
                    split = nonempty >= 2 && (mx > p->max_gapped_len || (big >= 2 && big < n));
                }
                if (split) np2.emplace_back(n);
                np2.back().push(pc.rec(i));
            }
        pieces.swap(np2);
    }
    // ---- gapped alignment of the inter-anchor intervals of this node (descriptors carry GLOBAL genome ids) ----
    struct GapRef { size_t piece, idx; bool dp; int64_t slot; };
    std::vector<GapRef> gaps; std::vector<DpSeqDesc> desc; int64_t n_dp = 0, code_total = 0;
    for (size_t q = 0; q < pieces.size(); q++) {
        const MatchVec &ch = pieces[q];
        for (size_t i = 0; i + 1 < ch.size(); i++) {
            int64_t tot = 0, mx = 0; int nonempty = 0; int64_t lo[MAUVE_MAX_SEQ], ln[MAUVE_MAX_SEQ]; bool rv[MAUVE_MAX_SEQ];
            for (int j = 0; j < n; j++) { gap_of(ch.rec(i), ch.rec(i + 1), j, lo[j], ln[j], rv[j]); tot += ln[j]; mx = std::max(mx, ln[j]); nonempty += ln[j] > 0; }
            if (!tot) continue;
            GapRef gr{q, i, false, -1};
            if (p->gapped && nonempty >= 2 && mx <= p->max_gapped_len) {
                gr.dp = true; gr.slot = n_dp++;
                for (int j = 0; j < n; j++) { DpSeqDesc d; d.genome = gm[j]; d.rev = rv[j]; d.lo0 = lo[j] - 1; d.len = ln[j]; desc.push_back(d); }
                code_total += tot;
            }
            gaps.push_back(gr);
        }
    }
    HIPCHK(c, c->pin_dcols.ensure(((size_t)code_total + 1) * sizeof(uint32_t)));
    uint32_t *dcols = c->pin_dcols.as<uint32_t>();
    std::vector<int64_t> dcol_off((size_t)n_dp + 1, 0), dscore((size_t)n_dp + 1, 0);
    int64_t cells = 0;
    c->dp_band_from = INT64_MAX;            // the progressive path splits long intervals instead
    const bool refine = p->refine_rounds > 0 && n >= 3 && n_dp > 0;
    std::vector<int64_t> dsp; if (refine) dsp.assign((size_t)n_dp + 1, 0);
    // DESIGN.md S13 (setRefinement): every interval of k >= 3 sequences is also aligned in the rotated orders r = 1 .. min(rounds, k-1) and keeps the
    // alignment whose sum-of-pairs score (dp_sp_scores) is highest, lowest rotation on ties.  A candidate's slots hold the rotated sequences, so its
    // column bits are slots of the rotation.  The candidates do not depend on the first alignment, only the choice does: ONE batch holds the
    // intervals and all their candidates (one sizing pass, one launch with more to balance), the columns stay on the device, the scores come back,
    // and only the winners' columns are compacted and copied out (dp_fetch_picked) -- at C4's root 57 k candidates of 29 k intervals, 1 k of them win.
    // Several contexts (mauve_set_shard) keep the two exchanged batches.
    struct Cand { int64_t iv; int r, k; uint8_t nz[MAUVE_MAX_SEQ]; };
    std::vector<Cand> cands;
    if (refine) {
        const double tr0 = now_ms();
        int64_t ccodes = 0;
        // (sized once and filled in place: 86 000 descriptor rows at C4's root; push_back by push_back this took 3 ms.  A batch large enough for the
        // device front gets only the intervals and the index of every interval's first candidate: the rotated rows are made on the device.)
        std::vector<DpSeqDesc> &all = c->prog_desc;
        std::vector<int32_t> cbase((size_t)n_dp + 1, 0);
        int64_t nc0 = 0;
        for (int64_t iv = 0; iv < n_dp; iv++) {
            int k = 0; for (int j = 0; j < n; j++) k += desc[(size_t)(iv * n + j)].len != 0;
            cbase[(size_t)iv] = (int32_t)nc0;
            if (k >= 3) nc0 += std::min(p->refine_rounds, k - 1);
        }
        cbase[(size_t)n_dp] = (int32_t)nc0;
        const bool rot_dev = !c->shard_on && nc0 > 0 && nc0 < (1LL << 30) && dp_desc_rotations_on_device(n_dp + nc0);
        if (!rot_dev) { all.resize((size_t)(n_dp + nc0) * n); std::copy(desc.begin(), desc.end(), all.begin()); }
        cands.resize((size_t)nc0);
        DpSeqDesc none; none.genome = gm[0]; none.rev = 0; none.lo0 = 0; none.len = 0;
        int64_t q0 = 0;
        for (int64_t iv = 0; iv < n_dp; iv++) {
            Cand cd; cd.iv = iv; cd.k = 0;
            int64_t tot = 0;
            const DpSeqDesc *di = &desc[(size_t)(iv * n)];
            for (int j = 0; j < n; j++) if (di[j].len) { cd.nz[cd.k++] = (uint8_t)j; tot += di[j].len; }
            if (cd.k < 3) continue;
            for (int r = 1; r <= p->refine_rounds && r < cd.k; r++, q0++) {
                cd.r = r; cands[(size_t)q0] = cd;
                if (!rot_dev) {
                    DpSeqDesc *o = &all[(size_t)(n_dp + q0) * n];
                    for (int j = 0; j < n; j++) o[j] = j < cd.k ? di[cd.nz[(j + r) % cd.k]] : none;
                }
                ccodes += tot;
            }
        }
        const int64_t nc = (int64_t)cands.size(), na = n_dp + nc;
        const bool one_batch = !c->shard_on;
        const double tr1 = now_ms();
        std::vector<int64_t> aoff((size_t)na + 1, 0), ascore((size_t)na + 1, 0), asp((size_t)na + 1, 0);
        std::vector<uint32_t> ccols;                             // (the exchanged form only)
        if (one_batch && rot_dev) {
            rc = dp_batch_run_desc_rot(c, n, n_dp, desc.data(), cbase.data(), na, &p->scoring, aoff.data(), ascore.data(), &cells, asp.data());
            if (rc) return rc;
        } else if (one_batch) {
            rc = dp_batch_run_desc(c, n, na, all.data(), &p->scoring, nullptr, aoff.data(), ascore.data(), &cells, false, asp.data());
            if (rc) return rc;
        } else {
            rc = dp_batch_run_desc(c, n, n_dp, all.data(), &p->scoring, dcols, aoff.data(), ascore.data(), &cells, true, asp.data());
            if (rc) return rc;
            if (nc) {
                ccols.resize((size_t)ccodes + 1);
                std::vector<int64_t> coff((size_t)nc + 1, 0);
                int64_t cells2 = 0;
                rc = dp_batch_run_desc(c, n, nc, all.data() + (size_t)n_dp * n, &p->scoring, ccols.data(), coff.data(), ascore.data() + n_dp, &cells2, true, asp.data() + n_dp);
                if (rc) return rc;
                cells += cells2;
                for (int64_t q = 0; q <= nc; q++) aoff[(size_t)(n_dp + q)] = aoff[(size_t)n_dp] + coff[(size_t)q];
            }
        }
        const double tr2 = now_ms();
        std::vector<int64_t> pick((size_t)n_dp);                // the batch entry whose alignment the interval keeps
        int64_t replaced = 0;
        for (int64_t iv = 0; iv < n_dp; iv++) { pick[(size_t)iv] = iv; dsp[(size_t)iv] = asp[(size_t)iv]; }
        for (int64_t q = 0; q < nc; q++) {                     // candidates of an interval are consecutive, rotations ascending
            const int64_t iv = cands[(size_t)q].iv;
            if (asp[(size_t)(n_dp + q)] > dsp[(size_t)iv]) { dsp[(size_t)iv] = asp[(size_t)(n_dp + q)]; replaced += pick[(size_t)iv] == iv; pick[(size_t)iv] = n_dp + q; }
        }
        if (one_batch) {
            rc = dp_fetch_picked(c, n_dp, pick.data(), aoff.data(), dcols, dcol_off.data());
            if (rc) return rc;
        } else {
            // (the originals are in dcols already; winners' columns are spliced in below)
            std::copy(aoff.begin(), aoff.begin() + n_dp + 1, dcol_off.begin());
            if (replaced) {
                std::vector<uint32_t> ncols; ncols.reserve((size_t)dcol_off[(size_t)n_dp]);
                std::vector<int64_t> noff((size_t)n_dp + 1, 0);
                for (int64_t iv = 0; iv < n_dp; iv++) {
                    noff[(size_t)iv] = (int64_t)ncols.size();
                    const int64_t q = pick[(size_t)iv];
                    if (q == iv) ncols.insert(ncols.end(), dcols + dcol_off[(size_t)iv], dcols + dcol_off[(size_t)iv + 1]);
                    else ncols.insert(ncols.end(), ccols.data() + (aoff[(size_t)q] - aoff[(size_t)n_dp]), ccols.data() + (aoff[(size_t)q + 1] - aoff[(size_t)n_dp]));
                }
                noff[(size_t)n_dp] = (int64_t)ncols.size();
                if ((int64_t)ncols.size() > code_total) { c->err = "refinement: more columns than bases"; return MAUVE_ERR_STATE; }
                memcpy(dcols, ncols.data(), ncols.size() * sizeof(uint32_t));
                dcol_off.swap(noff);
            }
        }
        if (dcol_off[(size_t)n_dp] > code_total) { c->err = "refinement: more columns than bases"; return MAUVE_ERR_STATE; }
        // winners that are rotations: their column bits are slots of the rotation -> the node's slots
        for (int64_t iv = 0; iv < n_dp; iv++) {
            const int64_t q = pick[(size_t)iv];
            dscore[(size_t)iv] = ascore[(size_t)q];
            if (q == iv) continue;
            const Cand &cd = cands[(size_t)(q - n_dp)];
            uint32_t map[MAUVE_MAX_SEQ];
            for (int j = 0; j < cd.k; j++) map[j] = 1u << cd.nz[(j + cd.r) % cd.k];
            for (int64_t k = dcol_off[(size_t)iv]; k < dcol_off[(size_t)iv + 1]; k++) {
                const uint32_t mc = dcols[(size_t)k]; uint32_t o = 0;
                for (int j = 0; j < cd.k; j++) if (mc >> j & 1u) o |= map[j];
                dcols[(size_t)k] = o;
            }
        }
        if (trace) fprintf(stderr, "[trace] node %d: refinement: %lld candidate alignments of %lld intervals, %lld replaced; candidates %.3f ms, batch %.3f, pick + fetch + remap %.3f\n", node,
                           (long long)nc, (long long)n_dp, (long long)replaced, tr1 - tr0, tr2 - tr1, now_ms() - tr2);
    } else {
        rc = dp_batch_run_desc(c, n, n_dp, desc.data(), &p->scoring, dcols, dcol_off.data(), dscore.data(), &cells, true, nullptr);   // (several contexts: the node's intervals are dealt out)
        if (rc) return rc;
    }
    P.n_gap_dp += n_dp; P.n_cells += cells;
    { const double tn4 = now_ms(); P.t_seed += tn1 - tn0; P.t_chain += tn2 - tn1; P.t_rec += tn3 - tn2; P.t_dp += tn4 - tn3; }
    if (trace) {
        int64_t big = 0; for (int64_t k = 0; k < n_dp; k++) { int64_t mx = 0; for (int j = 0; j < n; j++) mx = std::max(mx, desc[(size_t)(k * n + j)].len); big = std::max(big, mx); }
        fprintf(stderr, "[trace] node %d (n=%d, pool %lld): seed %.1f ms (%lld mums), chain %.1f, recursion %.1f, dp %.1f ms (%lld intervals, %lld cells, longest side %lld)\n",
                node, n, (long long)rest_len, tn1 - tn0, (long long)nm, tn2 - tn1, tn3 - tn2, now_ms() - tn3, (long long)n_dp, (long long)cells, (long long)big);
    }
    // ---- blocks ----
    const double tb0 = now_ms();
    uint32_t gfull = 0; for (int j = 0; j < n; j++) gfull |= 1u << gm[j];
    // local genome bits -> global ones, a byte at a time
    bool same_bits = true; for (int j = 0; j < n; j++) same_bits = same_bits && gm[j] == j;
    std::vector<std::array<uint32_t, 256>> lut(4);
    if (!same_bits)
        for (int by = 0; by < 4; by++)
            for (int v = 0; v < 256; v++) {
                uint32_t o = 0;
                for (int bit = 0; bit < 8; bit++) { const int j = by * 8 + bit; if ((v >> bit & 1) && j < n) o |= 1u << gm[j]; }
                lut[(size_t)by][(size_t)v] = o;
            }
    std::vector<std::vector<std::pair<int64_t, int64_t>>> placed((size_t)n);
    size_t gi = 0;
    const int N = P.N;
    for (size_t q = 0; q < pieces.size(); q++) {
        const MatchVec &ch = pieces[q];
        R.col_off.push_back((int64_t)R.cols.size());
        R.dp_score.push_back(0);
        P.n_anchor += (int64_t)ch.size(); P.n_multi++;
        for (size_t i = 0; i < ch.size(); i++) {
            R.cols.insert(R.cols.end(), (size_t)ch.len(i), gfull);
            if (gi < gaps.size() && gaps[gi].piece == q && gaps[gi].idx == i) {
                const GapRef &gr = gaps[gi++];
                if (gr.dp) {
                    const uint32_t *src = dcols + dcol_off[(size_t)gr.slot], *end = dcols + dcol_off[(size_t)gr.slot + 1];
                    if (same_bits) R.cols.insert(R.cols.end(), src, end);               // (the root, and any node of genomes 0 .. n-1)
                    else {
                        const size_t at = R.cols.size();
                        R.cols.resize(at + (size_t)(end - src));
                        uint32_t *dst = R.cols.data() + at;
                        for (; src < end; src++) { const uint32_t m = *src; *dst++ = lut[0][m & 255] | lut[1][m >> 8 & 255] | lut[2][m >> 16 & 255] | lut[3][m >> 24]; }
                    }
                    R.dp_score.back() += dscore[(size_t)gr.slot];
                } else {
                    for (int j = 0; j < n; j++) { int64_t lo, ln; bool rv; gap_of(ch.rec(i), ch.rec(i + 1), j, lo, ln, rv); R.cols.insert(R.cols.end(), (size_t)ln, 1u << gm[j]); }
                }
            }
        }
        const size_t base = R.iv_left.size();
        R.iv_left.resize(base + N, 0); R.iv_right.resize(base + N, 0); R.iv_reverse.resize(base + N, 0);
        const size_t last = ch.size() - 1;
        for (int j = 0; j < n; j++) {
            const int64_t s0 = ch.st(0)[j], s1 = ch.st(last)[j];
            int64_t le, re;
            if (s0 > 0) { le = s0; re = s1 + ch.len(last) - 1; } else { le = -s1; re = -s0 + ch.len(0) - 1; }
            R.iv_left[base + gm[j]] = le; R.iv_right[base + gm[j]] = re; R.iv_reverse[base + gm[j]] = s0 < 0;
            placed[(size_t)j].push_back({le, re});
        }
    }
    for (int j = 0; j < n; j++) for (const auto &pl : placed[(size_t)j]) P.rest[gm[j]].carve(pl.first, pl.second);
    P.t_blocks += now_ms() - tb0;
    if (trace) fprintf(stderr, "[trace] node %d: blocks %.1f ms\n", node, now_ms() - tb0);
    rc = prog_node(P, P.left[node]);
    if (rc) return rc;
    return prog_node(P, P.right[node]);
}

}  // namespace

extern "C" {

static int guide_tree_core(mauve_ctx *c, uint64_t pattern, int64_t *dist, int32_t *left, int32_t *right, int64_t bp_min_len, int64_t *bp);

// guide tree only (distance matrix in ppm, UPGMA merge order); dist may be NULL
int mauve_guide_tree(mauve_ctx *c, uint64_t pattern, int64_t *dist, int32_t *left, int32_t *right)
{
    return guide_tree_core(c, pattern, dist, left, right, -1, nullptr);
}

// the pairwise breakpoint estimate on its own (symmetric counts, bp[nseq*nseq])
int mauve_breakpoint_counts(mauve_ctx *c, uint64_t pattern, int64_t min_len, int64_t *bp)
{
    if (!c || !bp) return MAUVE_ERR_ARG;
    if (min_len < 0) { c->err = "breakpoint_counts: min_len must not be negative"; return MAUVE_ERR_ARG; }
    if (c->nseq < 2) { c->err = "breakpoint_counts: at least two genomes required"; return MAUVE_ERR_STATE; }
    std::vector<int32_t> l((size_t)(2 * c->nseq - 1)), r((size_t)(2 * c->nseq - 1));
    return guide_tree_core(c, pattern, nullptr, l.data(), r.data(), min_len, bp);
}

static int guide_tree_core(mauve_ctx *c, uint64_t pattern, int64_t *dist, int32_t *left, int32_t *right, int64_t bp_min_len, int64_t *bp)
{
    if (!c || !left || !right) return MAUVE_ERR_ARG;
    if (c->nseq < 2) { c->err = "guide_tree: at least two genomes required"; return MAUVE_ERR_STATE; }
    HIPCHK(c, hipSetDevice(c->device));
    { int rcm = materialize_tables(c); if (rcm) return rcm; }           // a resident result keeps its tables in buffers the passes below reuse
    const int N = c->nseq, M = 2 * N - 1;
    int64_t nm = 0;
    // similarity = sum of the pairwise match lengths per genome pair: summed on the device, the matches themselves stay there
    c->pair_sums_only = true; c->bp_min_len = bp ? bp_min_len : -1;
    int rc = seedpass_run(c, main_genome_set(c), pattern, MAUVE_MODE_PAIRWISE, 0, 1, nullptr, 0, &nm);
    c->pair_sums_only = false; c->bp_min_len = -1;
    if (rc) return rc;
    std::vector<int64_t> S((size_t)N * N, 0);
    if (c->shard_on) {
        // every rank ran the finder passes of its share of the genome pairs (seed_pass.hip): the sums of the others arrive here
        const size_t nn = (size_t)N * N;
        std::vector<int64_t> mine(2 * nn, 0);                 // length sums, then breakpoints
        if (c->pair_sums.size() == nn) std::copy(c->pair_sums.begin(), c->pair_sums.end(), mine.begin());
        if (bp && c->pair_bp.size() == nn) std::copy(c->pair_bp.begin(), c->pair_bp.end(), mine.begin() + (ptrdiff_t)nn);
        std::vector<std::pair<const char *, size_t>> parts;
        rc = shard_allgather(c, mine.data(), mine.size() * 8, parts);
        if (rc) return rc;
        c->pair_sums.assign(nn, 0); c->pair_bp.assign(nn, 0);
        for (const auto &pt : parts) {
            if (pt.second != mine.size() * 8) { c->err = "guide_tree: ranks disagree about the genome set"; return MAUVE_ERR_STATE; }
            const int64_t *v = reinterpret_cast<const int64_t *>(pt.first);
            for (size_t k = 0; k < nn; k++) { c->pair_sums[k] += v[k]; c->pair_bp[k] += v[nn + k]; }
        }
    }
    if (bp) {
        std::fill(bp, bp + (size_t)N * N, 0);
        if (c->pair_bp.size() == (size_t)N * N)
            for (int a = 0; a < N; a++) for (int b = a + 1; b < N; b++) bp[(size_t)a * N + b] = bp[(size_t)b * N + a] = c->pair_bp[(size_t)a * N + b];
    }
    if (c->pair_sums.size() == (size_t)N * N)
        for (int a = 0; a < N; a++) for (int b = a + 1; b < N; b++) { S[(size_t)a * N + b] = S[(size_t)b * N + a] = c->pair_sums[(size_t)a * N + b]; }
    std::vector<int64_t> D((size_t)M * M, 0), size((size_t)M, 0);
    std::vector<char> active((size_t)M, 0);
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) {
        const int64_t mn = std::min(c->lens[i], c->lens[j]); int64_t d = 1000000;
        if (i == j) d = 0;
        else if (mn > 0) d = 1000000 - std::min<int64_t>(1000000, S[(size_t)i * N + j] * 1000000 / mn);
        D[(size_t)i * M + j] = d; if (dist) dist[(size_t)i * N + j] = d;
    }
    for (int i = 0; i < N; i++) { active[i] = 1; size[i] = 1; left[i] = right[i] = -1; }
    for (int k = N; k < M; k++) {
        int ba = -1, bb = -1; int64_t bd = 0;
        for (int a = 0; a < k; a++) if (active[a]) for (int b = a + 1; b < k; b++) if (active[b])
            if (ba < 0 || D[(size_t)a * M + b] < bd) { ba = a; bb = b; bd = D[(size_t)a * M + b]; }
        left[k] = ba; right[k] = bb; size[k] = size[ba] + size[bb];
        for (int x = 0; x < k; x++) if (active[x] && x != ba && x != bb)
            D[(size_t)k * M + x] = D[(size_t)x * M + k] = (size[ba] * D[(size_t)ba * M + x] + size[bb] * D[(size_t)bb * M + x]) / size[k];
        active[ba] = active[bb] = 0; active[k] = 1;
    }
    return MAUVE_OK;
}

static int progressive_core(mauve_ctx *c, const mauve_params *p, mauve_align_sizes *sizes, const int32_t *given_left,
                            const int32_t *given_right, int32_t *tree_left, int32_t *tree_right, int64_t *dist);

int mauve_progressive_align(mauve_ctx *c, const mauve_params *p, mauve_align_sizes *sizes, int32_t *tree_left,
                            int32_t *tree_right, int64_t *dist)
{
    return progressive_core(c, p, sizes, nullptr, nullptr, tree_left, tree_right, dist);
}

// the caller's guide tree instead of the UPGMA one: leaves 0..N-1 childless, every internal node N..2N-2 with two
// distinct children of smaller id, every node but the root 2N-2 used exactly once
int mauve_progressive_align_tree(mauve_ctx *c, const mauve_params *p, mauve_align_sizes *sizes, const int32_t *tree_left,
                                 const int32_t *tree_right)
{
    if (!c || !tree_left || !tree_right) return MAUVE_ERR_ARG;
    const int N = c->nseq, M = 2 * N - 1;
    if (N < 2) { c->err = "progressive_align: at least two genomes required"; return MAUVE_ERR_STATE; }
    std::vector<char> used((size_t)M, 0);
    for (int k = 0; k < M; k++) {
        const int a = tree_left[k], b = tree_right[k];
        if (k < N) { if (a != -1 || b != -1) { c->err = "progressive_align_tree: a leaf has children"; return MAUVE_ERR_ARG; } continue; }
        if (a < 0 || b < 0 || a >= k || b >= k || a == b || used[(size_t)a] || used[(size_t)b]) {
            c->err = "progressive_align_tree: not a binary tree in merge order"; return MAUVE_ERR_ARG;
        }
        used[(size_t)a] = used[(size_t)b] = 1;
    }
    return progressive_core(c, p, sizes, tree_left, tree_right, nullptr, nullptr, nullptr);
}

}  // extern "C"

static int progressive_core(mauve_ctx *c, const mauve_params *p, mauve_align_sizes *sizes, const int32_t *given_left,
                            const int32_t *given_right, int32_t *tree_left, int32_t *tree_right, int64_t *dist)
{
    if (!c || !p || !sizes) return MAUVE_ERR_ARG;
    if (c->nseq < 2) { c->err = "progressive_align: at least two genomes required"; return MAUVE_ERR_STATE; }
    if (p->lcb_scoring != MAUVE_LCB_SCORE_LENGTH && p->lcb_scoring != MAUVE_LCB_SCORE_SP) { c->err = "progressive_align: unknown lcb_scoring"; return MAUVE_ERR_ARG; }
    if (p->seed_family && p->seed_pattern) { c->err = "progressive_align: seed_family takes its patterns from the weight, not from seed_pattern"; return MAUVE_ERR_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    c->rec_flags.clear();                 // (an earlier call that ended before its recursion may have left its flags behind)
    const double t0 = now_ms();
    const int N = c->nseq;
    int64_t sum = 0; for (int g = 0; g < N; g++) sum += c->lens[g];
    int w = p->seed_weight > 0 ? p->seed_weight : mauve_default_seed_weight(sum / N);
    uint64_t pat = p->seed_pattern ? p->seed_pattern : mauve_get_seed(w, p->seed_rank);
    if (!pat) { c->err = "progressive_align: no seed pattern for this weight/rank"; return MAUVE_ERR_ARG; }
    Prog P; P.c = c; P.p = p; P.N = N;
    P.left.assign((size_t)(2 * N - 1), -1); P.right.assign((size_t)(2 * N - 1), -1);
    int rc = 0;
    if (given_left) { std::copy(given_left, given_left + (2 * N - 1), P.left.begin()); std::copy(given_right, given_right + (2 * N - 1), P.right.begin()); }
    const bool need_bp = p->weight_scaling && p->bp_dist_scale_ppm > 0;
    const int64_t bp_min = p->bp_dist_min_score >= 0 ? p->bp_dist_min_score : 2 * (int64_t)mauve_seed_weight(pat);
    if (need_bp) P.bpd.assign((size_t)N * N, 0);
    bool have_bp = false;
    if (!given_left) { rc = guide_tree_core(c, pat, dist, P.left.data(), P.right.data(), bp_min, need_bp ? P.bpd.data() : nullptr); have_bp = need_bp; }
    if (rc) return rc;
    if (p->weight_scaling) {                // the distances come from the pairwise matches even when the tree is the caller's
        P.dist.assign((size_t)N * N, 0);
        if (given_left || !dist) {
            std::vector<int32_t> tl((size_t)(2 * N - 1)), tr((size_t)(2 * N - 1));
            rc = guide_tree_core(c, pat, P.dist.data(), tl.data(), tr.data(), bp_min, need_bp && !have_bp ? P.bpd.data() : nullptr);
            if (rc) return rc;
        } else std::copy(dist, dist + (size_t)N * N, P.dist.begin());
        if (need_bp) {                      // relative to the most rearranged pair
            int64_t mx = 0; for (int64_t v : P.bpd) mx = std::max(mx, v);
            for (int64_t &v : P.bpd) v = mx ? v * 1000000 / mx : 0;
        }
    }
    static const bool trace = getenv("MAUVE_TRACE") != nullptr;
    const double tg1 = now_ms();
    if (trace) fprintf(stderr, "[trace] progressive: guide tree %.1f ms\n", tg1 - t0);
    if (tree_left) std::copy(P.left.begin(), P.left.end(), tree_left);
    if (tree_right) std::copy(P.right.begin(), P.right.end(), tree_right);
    AlignResult &R = c->res;
    R.sz = mauve_align_sizes();
    R.mum_length.clear(); R.mum_start.clear(); R.lcb_left.clear(); R.lcb_right.clear(); R.lcb_weight.clear();
    R.anchor_length.clear(); R.anchor_start.clear(); R.anchor_lcb.clear(); R.iv_left.clear(); R.iv_right.clear();
    R.iv_reverse.clear(); R.col_off.clear(); R.cols.clear(); R.dp_score.clear();
    R.dev_pending = false; R.cols_pending = false; R.stale = false; R.genomes_replaced = false; R.dev_na = 0; R.dev_nm = 0; R.cols_ext = nullptr; R.cols_fill = 0; R.cols_dirty.clear();
    R.cols_fill = 0; R.cols_dirty.clear();           // mauve_align's prefilled-buffer invariant no longer holds
    P.R = &R;
    P.rest.assign((size_t)N, FreePool());
    for (int g = 0; g < N; g++) P.rest[(size_t)g].reset(c->lens[g]);
    rc = prog_node(P, 2 * N - 2);
    if (rc) return rc;
    const double tg2 = now_ms();
    if (trace) fprintf(stderr, "[trace] progressive: nodes %.1f ms\n", tg2 - tg1);
    const int64_t n_multi = P.n_multi;
    if (p->add_unaligned)
        for (int g = 0; g < N; g++)
            for (const auto &fr : P.rest[(size_t)g].free_) {
                const int64_t lo = fr.first, hi = fr.second;
                R.col_off.push_back((int64_t)R.cols.size());
                R.cols.insert(R.cols.end(), (size_t)(hi - lo + 1), 1u << g);
                for (int h = 0; h < N; h++) { R.iv_left.push_back(h == g ? lo : 0); R.iv_right.push_back(h == g ? hi : 0); R.iv_reverse.push_back(0); }
                R.dp_score.push_back(0);
            }
    R.col_off.push_back((int64_t)R.cols.size());
    // the multi-genome blocks double as the "LCB" table of the result (signed ends, weight unused)
    R.lcb_left.assign((size_t)n_multi * N, 0); R.lcb_right.assign((size_t)n_multi * N, 0); R.lcb_weight.assign((size_t)n_multi, 0);
    for (int64_t b = 0; b < n_multi; b++) for (int g = 0; g < N; g++) {
        const int64_t le = R.iv_left[(size_t)b * N + g], re = R.iv_right[(size_t)b * N + g];
        const bool rv = R.iv_reverse[(size_t)b * N + g] != 0;
        R.lcb_left[(size_t)b * N + g] = rv ? -le : le; R.lcb_right[(size_t)b * N + g] = rv ? -re : re;
    }
    R.sz.n_mums = 0; R.sz.n_lcb = n_multi; R.sz.n_anchor = 0; R.sz.n_iv = (int64_t)R.dp_score.size();
    R.n_cols = R.cols.size();
    R.sz.n_cols = (int64_t)R.cols.size(); R.sz.n_gap_dp = P.n_gap_dp; R.sz.n_dp_cells = P.n_cells;
    *sizes = R.sz;
    memset(&c->stage, 0, sizeof c->stage);
    c->stage.tree_ms = tg1 - t0; c->stage.seed_ms = P.t_seed; c->stage.chain_ms = P.t_chain; c->stage.recurse_ms = P.t_rec; c->stage.dp_ms = P.t_dp;
    c->stage.assemble_ms = P.t_blocks + (now_ms() - tg2);
    c->stage.total_ms = now_ms() - t0;
    return MAUVE_OK;
}
