// libMems/Aligner.h -- the chaining seam: mems::Aligner with its setters and align() as called at
// mauveAligner.cpp:668-698, and the LCB helpers the in-tree tools use on their own -- struct LCB,
// IdentifyBreakpoints, ComputeLCBs_v2, computeLCBAdjacencies_v2 (toGrimmFormat.cpp:51-79, projectAndStrip.cpp:110-112,
// sortContigs.cpp:55-84), EliminateOverlaps(MatchList&) (mauveAligner.cpp:596), transposeMatches (:634).
#ifndef MAUVE_HIP_ALIGNER_H
#define MAUVE_HIP_ALIGNER_H

#include "GappedAligner.h"
#include "IntervalList.h"
#include "MatchList.h"
#include <fstream>

namespace mems {

// LCB as toGrimmFormat.cpp:51-79 and sortContigs.cpp:80-83 read it: signed ends (negative = reverse), adjacencies per
// sequence with -1 = none (the -2 sentinel marks "not computed yet", toGrimmFormat.cpp:62), id and weight
static const int64 NO_ADJACENCY = -1;
static const int64 ADJACENCY_UNSET = -2;
struct LCB {
    std::vector<int64> left_end, right_end;
    std::vector<int64> left_adjacency, right_adjacency;
    int lcb_id;
    double weight;
    LCB() : lcb_id(0), weight(0) {}
};

// EliminateOverlaps (mauveAligner.cpp:596): matches cropped / dropped until no two overlap in any sequence
inline void EliminateOverlaps(MatchList &ml)
{
    if (ml.empty()) return;
    const uint N = ml[0]->SeqCount();
    int64_t n = (int64_t)ml.size();
    std::vector<int64_t> len((size_t)n), st((size_t)n * N);
    for (int64_t i = 0; i < n; i++) {
        len[(size_t)i] = (int64_t)ml[(size_t)i]->Length();
        for (uint g = 0; g < N; g++) st[(size_t)i * N + g] = ml[(size_t)i]->Start(g);
    }
    if (mauve_eliminate_overlaps((int)N, &n, len.data(), st.data()) != MAUVE_OK) throw genome::gnException("EliminateOverlaps: every match must be defined in every sequence");
    for (size_t i = (size_t)n; i < ml.size(); i++) ml[i]->Free();
    ml.resize((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        ml[(size_t)i]->SetLength((gnSeqI)len[(size_t)i]);
        for (uint g = 0; g < N; g++) ml[(size_t)i]->SetStart(g, st[(size_t)i * N + g]);
    }
}

// breakpoints of an N-way, overlap-free match list: indices (genome-0 order) after which the collinear run ends
inline void IdentifyBreakpoints(MatchList &ml, std::vector<gnSeqI> &breakpoints)
{
    breakpoints.clear();
    if (ml.empty()) return;
    const uint N = ml[0]->SeqCount(); const int64_t n = (int64_t)ml.size();
    std::sort(ml.begin(), ml.end(), MatchStartComparator<Match>(0));
    std::vector<int64_t> len((size_t)n), st((size_t)n * N), lcb((size_t)n), le((size_t)n * N), re((size_t)n * N), wt((size_t)n), la((size_t)n * N), ra((size_t)n * N);
    for (int64_t i = 0; i < n; i++) { len[(size_t)i] = (int64_t)ml[(size_t)i]->Length(); for (uint g = 0; g < N; g++) st[(size_t)i * N + g] = ml[(size_t)i]->Start(g); }
    int64_t K = 0;
    if (mauve_lcb_chain((int)N, n, len.data(), st.data(), 0, 0, lcb.data(), &K, le.data(), re.data(), wt.data(), la.data(), ra.data()) != MAUVE_OK)
        throw genome::gnException("IdentifyBreakpoints: N-way matches required");
    for (int64_t i = 0; i < n; i++) if (i + 1 == n || lcb[(size_t)i] != lcb[(size_t)i + 1]) breakpoints.push_back((gnSeqI)i);
}

// LCBs between consecutive breakpoints with their weights (sum of match lengths * sequences)
inline void ComputeLCBs_v2(MatchList &ml, const std::vector<gnSeqI> &breakpoints, std::vector<MatchList> &lcb_list, std::vector<int64> &weights)
{
    lcb_list.clear(); weights.clear();
    size_t cur = 0;
    for (gnSeqI bp : breakpoints) {
        MatchList l; l.seq_table = ml.seq_table; l.seq_filename = ml.seq_filename;
        int64 w = 0;
        for (; cur <= (size_t)bp && cur < ml.size(); cur++) { l.push_back(ml[cur]); w += (int64)ml[cur]->Length() * (int64)ml[cur]->SeqCount(); }
        lcb_list.push_back(l); weights.push_back(w);
    }
}

// adjacency table of the LCBs (toGrimmFormat.cpp:51-79): ends signed by orientation, neighbours by left end per sequence
inline void computeLCBAdjacencies_v2(std::vector<MatchList> &lcb_list, const std::vector<int64> &weights, std::vector<LCB> &adjacencies)
{
    const size_t K = lcb_list.size();
    adjacencies.assign(K, LCB());
    if (!K) return;
    const uint N = lcb_list[0].empty() ? 0 : lcb_list[0][0]->SeqCount();
    for (size_t l = 0; l < K; l++) {
        LCB &b = adjacencies[l];
        b.lcb_id = (int)l; b.weight = l < weights.size() ? (double)weights[l] : 0;
        b.left_end.assign(N, 0); b.right_end.assign(N, 0);
        b.left_adjacency.assign(N, ADJACENCY_UNSET); b.right_adjacency.assign(N, ADJACENCY_UNSET);
        for (const Match *m : lcb_list[l])
            for (uint g = 0; g < N; g++) {
                if (m->Start(g) == NO_MATCH) continue;
                const int64 le = (int64)m->LeftEnd(g), re = (int64)m->RightEnd(g);
                const bool rev = m->Start(g) < 0;
                if (!b.left_end[g] || le < std::llabs(b.left_end[g])) b.left_end[g] = rev ? -le : le;
                if (!b.right_end[g] || re > std::llabs(b.right_end[g])) b.right_end[g] = rev ? -re : re;
            }
    }
    std::vector<size_t> idx(K);
    for (uint g = 0; g < N; g++) {
        for (size_t l = 0; l < K; l++) idx[l] = l;
        std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return std::llabs(adjacencies[a].left_end[g]) < std::llabs(adjacencies[b].left_end[g]); });
        for (size_t r = 0; r < K; r++) {
            adjacencies[idx[r]].left_adjacency[g] = r > 0 ? (int64)idx[r - 1] : NO_ADJACENCY;
            adjacencies[idx[r]].right_adjacency[g] = r + 1 < K ? (int64)idx[r + 1] : NO_ADJACENCY;
        }
    }
}

// the same table over the intervals of an alignment, each interval one LCB (toGrimmFormat.cpp:54, sortContigs.cpp:58)
inline void computeLCBAdjacencies_v2(IntervalList &iv_list, const std::vector<int64> &weights, std::vector<LCB> &adjacencies)
{
    const size_t K = iv_list.size();
    adjacencies.assign(K, LCB());
    if (!K) return;
    uint N = (uint)iv_list.seq_table.size();
    for (const Interval &iv : iv_list) N = std::max(N, iv.SeqCount());
    for (size_t l = 0; l < K; l++) {
        LCB &b = adjacencies[l]; const Interval &iv = iv_list[l];
        b.lcb_id = (int)l; b.weight = l < weights.size() ? (double)weights[l] : 0;
        b.left_end.assign(N, 0); b.right_end.assign(N, 0);
        b.left_adjacency.assign(N, ADJACENCY_UNSET); b.right_adjacency.assign(N, ADJACENCY_UNSET);
        for (uint g = 0; g < iv.SeqCount(); g++) {
            if (!iv.LeftEnd(g)) continue;
            const bool rev = iv.Orientation(g) == AbstractMatch::reverse;
            b.left_end[g] = rev ? -(int64)iv.LeftEnd(g) : (int64)iv.LeftEnd(g); b.right_end[g] = rev ? -(int64)iv.RightEnd(g) : (int64)iv.RightEnd(g);
        }
    }
    std::vector<size_t> idx;
    for (uint g = 0; g < N; g++) {
        idx.clear();
        for (size_t l = 0; l < K; l++) if (adjacencies[l].left_end[g]) idx.push_back(l);      // intervals without the sequence keep the -2 sentinel
        std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return std::llabs(adjacencies[a].left_end[g]) < std::llabs(adjacencies[b].left_end[g]); });
        for (size_t r = 0; r < idx.size(); r++) {
            adjacencies[idx[r]].left_adjacency[g] = r > 0 ? (int64)idx[r - 1] : NO_ADJACENCY;
            adjacencies[idx[r]].right_adjacency[g] = r + 1 < idx.size() ? (int64)idx[r + 1] : NO_ADJACENCY;
        }
    }
}

// mauveAligner.cpp:634: the starts of every match in sequence seqI, in list order
inline void transposeMatches(MatchList &ml, uint seqI, std::vector<int64> &starts)
{
    starts.clear();
    for (const Match *m : ml) starts.push_back(m->Start(seqI));
}

// One set of LCBs as a signed permutation per sequence: the LCB ids (1-based, numbered along sequence 0) in the order
// their left ends occur in that sequence, negative where the LCB lies reversed -- the rows toGrimmFormat.cpp:58-77 prints,
// tab-separated, one line per sequence, a blank line after the set.
inline void WritePermutation(std::ostream &os, uint N, int64_t K, const std::vector<int64_t> &left_end, const std::vector<int64_t> &left_adj, const std::vector<int64_t> &right_adj)
{
    for (uint g = 0; g < N; g++) {
        int64_t l = 0;
        while (l < K && left_adj[(size_t)l * N + g] != NO_ADJACENCY) l++;
        for (bool first = true; l >= 0 && l < K; l = right_adj[(size_t)l * N + g], first = false) {
            if (!first) os << '\t';
            if (left_end[(size_t)l * N + g] < 0) os << '-';
            os << l + 1;
        }
        os << '\n';
    }
    os << '\n';
}

class Aligner {
public:
    explicit Aligner(uint seq_count) : seq_count_(seq_count), gal_(nullptr) { mauve_default_params(&p_); }
    void SetMinRecursionGapLength(gnSeqI n) { p_.min_recursive_gap = (int64_t)n; }     // :670-672
    void SetGappedAligner(GappedAligner &ga) { gal_ = &ga; }                            // :674
    void SetMaxGappedAlignmentLength(gnSeqI n) { p_.max_gapped_len = (int64_t)n; }     // :675-676
    // no reference counterpart: gaps above the limit and up to n go through the banded DP instead of staying unaligned
    void SetMaxBandedAlignmentLength(gnSeqI n) { p_.max_banded_len = (int64_t)n; }
    void SetMaxExtensionIterations(uint n) { p_.max_extension_iters = (int32_t)n; }      // :687-690 (LCB extension)
    void SetSeedPattern(int64 seed) { p_.seed_pattern = (uint64_t)seed; }
    // :678-686 (--permutation-matrix-output / --permutation-matrix-min-weight; weight already x seq_count): align() writes
    // a signed permutation for every set of LCBs the greedy breakpoint elimination passes through between this minimum
    // weight and LCB_size -- the set at min_weight first, then one per lightest LCB removed, the set at LCB_size last.
    void SetPermutationOutput(const std::string &filename, int64 min_weight) { perm_fn_ = filename; perm_weight_ = min_weight; }
    void SetScoring(const PairwiseScoringScheme &pss)
    {
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) p_.scoring.matrix[i][j] = pss.matrix[i][j];
        p_.scoring.gap_open = pss.gap_open; p_.scoring.gap_extend = pss.gap_extend;
    }
    // align(match_list, interval_list, 0, LCB_size, recursive, extend_lcbs, gapped, tree_filename)  (:698).
    // LCB_size < 0 = collinear hack (:665-666).  The matches of match_list are what gets chained: they go to the device
    // as they are (mauve_align_matches); only the recursion and the LCB extension search for further anchors.  With
    // the built-in HipGappedAligner (or none) installed, all inter-anchor intervals are aligned in one batch of the DP
    // kernel; any other GappedAligner is called once per interval with the two flanking anchors (MatchRecord.h:311).
    void align(MatchList &ml, IntervalList &il, double, int64 LCB_size, boolean recursive, boolean extend_lcbs, boolean gapped, std::string = "")
    {
        HipContext &hc = HipContext::global();
        if (ml.seq_table.size() != seq_count_) throw genome::gnException("Aligner::align: sequence count mismatch");
        ml.upload(hc);
        mauve_params p = p_;
        if (!p.seed_pattern) p.seed_pattern = (uint64_t)ml.seed_pattern;
        p.collinear = LCB_size < 0; p.lcb_weight = LCB_size < 0 ? -1 : LCB_size;
        p.recursive = recursive; p.gapped = gapped; p.extend_lcbs = extend_lcbs;
        const uint N = seq_count_;
        std::vector<int64_t> len(ml.size()), st(ml.size() * N, 0);
        for (size_t i = 0; i < ml.size(); i++) {
            len[i] = (int64_t)ml[i]->Length();
            for (uint g = 0; g < N && g < ml[i]->SeqCount(); g++) st[i * N + g] = ml[i]->Start(g);
        }
        if (!perm_fn_.empty() && perm_weight_ >= 0 && LCB_size >= 0) write_permutations(len, st, LCB_size);
        const bool foreign = gapped && gal_ && dynamic_cast<HipGappedAligner *>(gal_) == nullptr;
        if (!foreign)
            hc.check(mauve_align_matches(hc.get(), &p, (int64_t)ml.size(), len.data(), st.data(), &il.sizes), "mauve_align_matches");
        else {
            int64_t n_dp = 0, n_codes = 0;
            hc.check(mauve_align_begin_matches(hc.get(), &p, (int64_t)ml.size(), len.data(), st.data(), &n_dp, &n_codes), "mauve_align_begin_matches");
            std::vector<int64_t> left((size_t)n_dp * (1 + N)), right((size_t)n_dp * (1 + N));
            hc.check(mauve_align_dp_anchors(hc.get(), left.data(), right.data()), "mauve_align_dp_anchors");
            std::vector<uint32_t> cols; std::vector<int64_t> col_off(1, 0), score((size_t)n_dp, 0);
            gal_->SetMaxAlignmentLength((gnSeqI)p.max_gapped_len);
            for (int64_t k = 0; k < n_dp; k++) {
                Match l(N), r(N);
                l.SetLength((gnSeqI)left[(size_t)k * (1 + N)]); r.SetLength((gnSeqI)right[(size_t)k * (1 + N)]);
                for (uint g = 0; g < N; g++) { l.SetStart(g, left[(size_t)k * (1 + N) + 1 + g]); r.SetStart(g, right[(size_t)k * (1 + N) + 1 + g]); }
                GappedAlignment cr;
                const bool ok = gal_->Align(cr, &l, &r, ml.seq_table);
                if (ok) {
                    const std::vector<std::string> &rows = cr.GetAlignment();
                    for (gnSeqI c = 0; c < cr.AlignmentLength(); c++) {
                        uint32_t mask = 0;
                        for (uint g = 0; g < N && g < rows.size(); g++) if (rows[g][(size_t)c] != '-') mask |= 1u << g;
                        cols.push_back(mask);
                    }
                } else {                                          // the plug declined: the bases go out unaligned, sequence by sequence
                    std::string s; int64 st0;
                    for (uint g = 0; g < N; g++) { getInterveningSequence(&l, &r, g, ml.seq_table, s, st0); cols.insert(cols.end(), s.size(), 1u << g); }
                }
                col_off.push_back((int64_t)cols.size());
            }
            if (cols.empty()) cols.push_back(0);
            hc.check(mauve_align_finish(hc.get(), cols.data(), col_off.data(), score.data(), 0, &il.sizes), "mauve_align_finish");
        }
        il.seq_table = ml.seq_table; il.seq_filename = ml.seq_filename;
        il.fetch(hc, seq_count_);
    }
    // Aligner::align resumed from LCBs (mauveAligner.cpp:705-722: an IntervalList read back with --lcb-input; :723-744 --realign-lcb
    // hands the matches of each LCB to align() again): every interval of `lcbs` that holds N-way Matches is one chain -- its matches
    // stay together, no overlap / breakpoint elimination, no LCB extension (the call site passes false) -- and goes through recursive
    // anchoring and the gapped alignment of its inter-anchor intervals on the device (mauve_align_lcbs).  Intervals without such
    // matches (already aligned blocks, single-genome islands) are skipped, as the call site's dynamic_cast skips them.
    void realign(IntervalList &lcbs, IntervalList &il, boolean recursive, boolean gapped)
    {
        HipContext &hc = HipContext::global();
        if (lcbs.seq_table.size() != seq_count_) throw genome::gnException("Aligner::realign: sequence count mismatch");
        MatchList ml; ml.seq_table = lcbs.seq_table; ml.seq_filename = lcbs.seq_filename;
        ml.upload(hc);
        const uint N = seq_count_;
        std::vector<int64_t> len, st, id;
        int64_t nl = 0;
        for (size_t iv = 0; iv < lcbs.size(); iv++) {
            bool any = false;
            for (AbstractMatch *am : lcbs[iv].GetMatches()) {
                Match *m = dynamic_cast<Match *>(am);
                if (!m || m->Multiplicity() < N) continue;
                Match mm(*m);
                if (mm.Start(0) < 0) mm.Invert();               // anchors are forward in genome 0
                len.push_back((int64_t)mm.Length()); id.push_back(nl); any = true;
                for (uint g = 0; g < N; g++) st.push_back(mm.Start(g));
            }
            if (any) nl++;
        }
        mauve_params p = p_;
        p.recursive = recursive; p.gapped = gapped; p.extend_lcbs = 0;
        hc.check(mauve_align_lcbs(hc.get(), &p, (int64_t)len.size(), len.data(), st.data(), id.data(), &il.sizes), "mauve_align_lcbs");
        il.seq_table = lcbs.seq_table; il.seq_filename = lcbs.seq_filename;
        il.fetch(hc, seq_count_);
    }
private:
    // the LCB sets of the N-way matches between perm_weight_ and LCB_size (the chain align() runs, before recursion)
    void write_permutations(const std::vector<int64_t> &len_in, const std::vector<int64_t> &st_in, int64 LCB_size)
    {
        const uint N = seq_count_;
        std::vector<int64_t> len, st;
        for (size_t i = 0; i < len_in.size(); i++) {
            bool all = true;
            for (uint g = 0; g < N; g++) all = all && st_in[i * N + g] != 0;
            if (!all) continue;
            len.push_back(len_in[i]); st.insert(st.end(), st_in.begin() + (std::ptrdiff_t)(i * N), st_in.begin() + (std::ptrdiff_t)((i + 1) * N));
        }
        int64_t n = (int64_t)len.size();
        std::ofstream out(perm_fn_.c_str());
        if (!out) throw genome::gnException("Aligner::align: cannot write the permutation file " + perm_fn_);
        if (!n) return;
        if (mauve_eliminate_overlaps((int)N, &n, len.data(), st.data()) != MAUVE_OK) throw genome::gnException("Aligner::align: permutation output: bad match list");
        std::vector<int64_t> lcb((size_t)n), le((size_t)n * N), re((size_t)n * N), wt((size_t)n), la((size_t)n * N), ra((size_t)n * N);
        for (int64_t w = std::min<int64_t>(perm_weight_, LCB_size);;) {
            int64_t K = 0;
            if (mauve_lcb_chain((int)N, n, len.data(), st.data(), w, 0, lcb.data(), &K, le.data(), re.data(), wt.data(), la.data(), ra.data()) != MAUVE_OK)
                throw genome::gnException("Aligner::align: permutation output: chaining failed");
            WritePermutation(out, N, K, le, la, ra);
            if (w >= LCB_size || K <= 1) break;
            int64_t lightest = wt[0];
            for (int64_t l = 1; l < K; l++) lightest = std::min(lightest, wt[(size_t)l]);
            w = std::min<int64_t>(std::max(w, lightest) + 1, LCB_size);
        }
    }
    uint seq_count_;
    mauve_params p_;
    GappedAligner *gal_;
    std::string perm_fn_;
    int64 perm_weight_ = -1;
};

}  // namespace mems
#endif
