// libMems/Backbone.h -- the backbone stage of progressiveMauve as its call site uses it (applyBackbone,
// progressiveMauve.cpp:226-260) and the .backbone / .bbcols files the in-tree tools read (bbFilter.cpp:75-90,
// bbAnalyze.cpp:992-1015, backbone_global_to_local.cpp:33-34, bbBreakOnGenes.cpp:310-354, getOrthologList.cpp:97).
// detectBackbone runs on the device (mauve_backbone_alignment, DESIGN.md S12: the BigGapsDetector rule); the list
// helpers work on a few thousand rows and are host code.
//   .backbone : header "seq0_leftend\tseq0_rightend\tseq1_leftend..." then one row per segment, two signed numbers per
//               sequence (negative = reverse strand, 0 0 = not in the segment)          (bbFilter.cpp:28-31 reads them so)
//   .bbcols   : one row per segment: interval, first column, columns, then the sequences in it (bbAnalyze.cpp:1010-1012)
// The homology HMM behind detectAndApplyBackbone (HomologyHMM, not in the reference tree) has a frozen form (DESIGN.md S12b:
// a two-state Viterbi path per interval and genome pair, mauve_apply_homology_alignment): residues the path classes as unrelated
// to every other genome of their column leave it; the backbone is the big-gaps one of the alignment after that.
#ifndef MAUVE_HIP_BACKBONE_H
#define MAUVE_HIP_BACKBONE_H

#include "IntervalList.h"
#include "dynamic_bitset.h"
#include <cmath>
#include <algorithm>
#include <cstdlib>
#include <sstream>

namespace mems {

typedef std::vector<std::pair<int64, int64>> bb_seqentry_t;               // per sequence: signed left end, signed right end
typedef std::pair<size_t, std::vector<size_t>> bb_colentry_t;             // interval; first column, columns, sequences...
struct bb_entry_t { bb_seqentry_t bb_seq; std::vector<size_t> bb_cols; size_t iv; bb_entry_t() : iv(0) {} };

// one backbone segment of one interval
struct BackboneSegment {
    size_t iv; gnSeqI left_col, length; uint32_t genomes;
    bb_seqentry_t ends;                                                   // signed ends per sequence
    uint SeqCount() const { return (uint)ends.size(); }
    uint Multiplicity() const { uint n = 0; for (uint g = 0; g < ends.size(); g++) n += (genomes >> g & 1); return n; }
    int64 Start(uint g) const { return ends[g].first; }
    gnSeqI LeftEnd(uint g) const { return (gnSeqI)std::llabs(ends[g].first); }
    gnSeqI RightEnd(uint g) const { return (gnSeqI)std::llabs(ends[g].second); }
    gnSeqI Length(uint g) const { return ends[g].first ? RightEnd(g) - LeftEnd(g) + 1 : 0; }
};
typedef std::vector<std::vector<BackboneSegment>> backbone_list_t;        // per interval of the list

// the detector plug of detectBackbone (progressiveMauve.cpp:242-243); BigGapsDetector is the one the device implements
class HssDetector { public: virtual ~HssDetector() {} };
class BigGapsDetector : public HssDetector {
public:
    explicit BigGapsDetector(size_t big_gap_size) : gap_(big_gap_size) {}
    size_t gapSize() const { return gap_; }
private:
    size_t gap_;
};

// islands of the last detectBackbone-style call: interval, the pair, who has the residues, columns, signed ends
struct PairIsland { size_t iv; uint seq_a, seq_b, who; gnSeqI left_col, right_col; int64 left_end, right_end; };

inline void detectBackboneAndIslands(IntervalList &il, size_t gap, backbone_list_t &bb_list, std::vector<PairIsland> *islands)
{
    HipContext &hc = HipContext::global();
    const size_t K = il.size();
    uint N = (uint)il.seq_table.size();
    for (const Interval &iv : il) N = std::max(N, iv.SeqCount());
    std::vector<int64_t> left(K * N, 0), right(K * N, 0), col_off(K + 1, 0);
    std::vector<int8_t> rev(K * N, 0);
    std::vector<uint32_t> cols;
    for (size_t i = 0; i < K; i++) {
        const Interval &iv = il[i];
        for (uint g = 0; g < iv.SeqCount(); g++) {
            left[i * N + g] = (int64_t)iv.LeftEnd(g); right[i * N + g] = iv.LeftEnd(g) ? (int64_t)iv.RightEnd(g) : 0;
            rev[i * N + g] = iv.LeftEnd(g) && iv.Orientation(g) == AbstractMatch::reverse;
        }
        cols.insert(cols.end(), iv.Columns().begin(), iv.Columns().end());
        col_off[i + 1] = (int64_t)cols.size();
    }
    if (cols.empty()) cols.push_back(0);
    int64_t n_seg = 0, n_isl = 0;
    hc.check(mauve_backbone_alignment(hc.get(), (int)N, (int64_t)K, left.data(), right.data(), rev.data(), col_off.data(), cols.data(), (int64_t)gap, &n_seg, &n_isl),
             "mauve_backbone_alignment");
    std::vector<int64_t> s_iv((size_t)n_seg), s_col((size_t)n_seg), s_len((size_t)n_seg), s_l((size_t)n_seg * N), s_r((size_t)n_seg * N), isl((size_t)n_isl * 8);
    std::vector<uint32_t> s_mask((size_t)n_seg);
    hc.check(mauve_backbone_fetch(hc.get(), s_iv.data(), s_col.data(), s_len.data(), s_mask.data(), s_l.data(), s_r.data(), isl.data()), "mauve_backbone_fetch");
    bb_list.assign(K, std::vector<BackboneSegment>());
    for (size_t s = 0; s < (size_t)n_seg; s++) {
        BackboneSegment b; b.iv = (size_t)s_iv[s]; b.left_col = (gnSeqI)s_col[s]; b.length = (gnSeqI)s_len[s]; b.genomes = s_mask[s];
        b.ends.resize(N);
        for (uint g = 0; g < N; g++) b.ends[g] = std::make_pair((int64)s_l[s * N + g], (int64)s_r[s * N + g]);
        bb_list[b.iv].push_back(b);
    }
    if (islands) {
        islands->clear();
        for (size_t k = 0; k < (size_t)n_isl; k++) {
            const int64_t *r = &isl[k * 8];
            islands->push_back(PairIsland{(size_t)r[0], (uint)r[1], (uint)r[2], (uint)r[3], (gnSeqI)r[4], (gnSeqI)r[5], (int64)r[6], (int64)r[7]});
        }
    }
}

// detectBackbone(iv_list, bb_list, &bgd)  (progressiveMauve.cpp:242-243)
inline void detectBackbone(IntervalList &il, backbone_list_t &bb_list, const HssDetector *detector)
{
    const BigGapsDetector *bgd = dynamic_cast<const BigGapsDetector *>(detector);
    if (!bgd) throw genome::gnException("detectBackbone: only BigGapsDetector is implemented on the device");
    detectBackboneAndIslands(il, bgd->gapSize(), bb_list, nullptr);
}

// ---- the homology HMM's knobs (progressiveMauve.cpp:231-237).  The GC adaptation of libMems' emission matrix is not reproduced
// (the frozen form scores match / mismatch from the identity alone); gc is carried. ----
struct Params {
    double iGoHomologous, iGoUnrelated, identity, gc;
    Params() : iGoHomologous(0.00001), iGoUnrelated(0.000000001), identity(0.7), gc(0.5) {}
};
inline double computeGC(const std::vector<genome::gnSequence *> &seq_table)
{
    double gc = 0, all = 0;
    for (const genome::gnSequence *s : seq_table) {
        const std::string t = s->ToString();
        for (char ch : t) { const char u = (char)toupper((unsigned char)ch); if (u == 'G' || u == 'C') gc += 1; if (u == 'A' || u == 'C' || u == 'G' || u == 'T') all += 1; }
    }
    return all > 0 ? gc / all : 0.5;
}
inline Params getAdaptedHoxdMatrixParameters(double gc_content) { Params p; p.gc = gc_content; return p; }
inline void adaptToPercentIdentity(Params &p, double identity) { p.identity = identity; }
inline void detectAndApplyBackbone(IntervalList &il, backbone_list_t &bb_list, const Params &hp)
{
    HipContext &hc = HipContext::global();
    const size_t K = il.size();
    const uint N = (uint)il.seq_table.size();
    if (K && N) {
        {   // the interval list refers to its own sequence table: those are the genomes the pass compares.  Uploaded like a MatchList's
            // (contig starts and ambiguity bitmaps go along, so that a later search on this context sees them; the homology pass itself
            // reads an ambiguous base as A)
            MatchList tmp; tmp.seq_table = il.seq_table;
            tmp.upload(hc);
        }
        std::vector<int64_t> left(K * N, 0), right(K * N, 0), col_off(K + 1, 0), noff(K + 1, 0);
        std::vector<int8_t> rev(K * N, 0);
        std::vector<uint32_t> cols; size_t residues = 0;
        for (size_t i = 0; i < K; i++) {
            const Interval &iv = il[i];
            for (uint g = 0; g < iv.SeqCount() && g < N; g++) {
                left[i * N + g] = (int64_t)iv.LeftEnd(g); right[i * N + g] = iv.LeftEnd(g) ? (int64_t)iv.RightEnd(g) : 0;
                rev[i * N + g] = iv.LeftEnd(g) && iv.Orientation(g) == AbstractMatch::reverse;
                if (iv.LeftEnd(g)) residues += (size_t)(right[i * N + g] - left[i * N + g] + 1);
            }
            cols.insert(cols.end(), iv.Columns().begin(), iv.Columns().end());
            col_off[i + 1] = (int64_t)cols.size();
        }
        if (!cols.empty()) {
            mauve_hmm_params h;
            mauve_hmm_params_from(hp.identity, hp.iGoHomologous, hp.iGoUnrelated, &h);
            std::vector<uint32_t> ncols(std::max(residues, cols.size()) + 1);
            int64_t moved = 0;
            hc.check(mauve_apply_homology_alignment(hc.get(), (int)N, (int64_t)K, left.data(), right.data(), rev.data(), col_off.data(), cols.data(), &h, noff.data(), ncols.data(), &moved),
                     "mauve_apply_homology_alignment");
            if (moved)
                for (size_t i = 0; i < K; i++) {
                    std::vector<int64> l(left.begin() + i * N, left.begin() + (i + 1) * N), r(right.begin() + i * N, right.begin() + (i + 1) * N);
                    std::vector<char> rv(rev.begin() + i * N, rev.begin() + (i + 1) * N);
                    il[i] = Interval(l, r, rv, std::vector<uint32_t>(ncols.begin() + noff[i], ncols.begin() + noff[i + 1]));
                }
        }
    }
    BigGapsDetector bgd(20);
    detectBackbone(il, bb_list, &bgd);
}

// ---- .backbone (sequence coordinates) ----
inline void writeBackboneSeqFile(std::ostream &os, const std::vector<bb_seqentry_t> &rows)
{
    const size_t N = rows.empty() ? 0 : rows[0].size();
    for (size_t g = 0; g < N; g++) os << (g ? "\t" : "") << "seq" << g << "_leftend\tseq" << g << "_rightend";
    os << '\n';
    for (const bb_seqentry_t &r : rows) {
        for (size_t g = 0; g < r.size(); g++) os << (g ? "\t" : "") << r[g].first << '\t' << r[g].second;
        os << '\n';
    }
}
inline void writeBackboneSeqCoordinates(const backbone_list_t &bb_list, const IntervalList &, std::ostream &os)
{
    std::vector<bb_seqentry_t> rows;
    for (const std::vector<BackboneSegment> &v : bb_list) for (const BackboneSegment &b : v) rows.push_back(b.ends);
    writeBackboneSeqFile(os, rows);
}
inline void readBackboneSeqFile(std::istream &is, std::vector<bb_seqentry_t> &rows)
{
    rows.clear();
    std::string line;
    if (!std::getline(is, line)) return;
    if (line.compare(0, 3, "seq") != 0) throw genome::gnException("readBackboneSeqFile: not a backbone file (header missing)");
    while (std::getline(is, line)) {
        if (line.empty()) continue;
        std::istringstream ls(line); bb_seqentry_t r; int64 a, b;
        while (ls >> a >> b) r.push_back(std::make_pair(a, b));
        if (!rows.empty() && r.size() != rows[0].size()) throw genome::gnException("readBackboneSeqFile: ragged row");
        rows.push_back(r);
    }
}
// ---- .bbcols (alignment columns) ----
inline void writeBackboneColumns(std::ostream &os, const backbone_list_t &bb_list)
{
    for (const std::vector<BackboneSegment> &v : bb_list)
        for (const BackboneSegment &b : v) {
            os << b.iv << '\t' << b.left_col << '\t' << b.length;
            for (uint g = 0; g < b.SeqCount(); g++) if (b.genomes >> g & 1) os << '\t' << g;
            os << '\n';
        }
}
inline void readBackboneColsFile(std::istream &is, std::vector<bb_colentry_t> &rows)
{
    rows.clear();
    std::string line;
    while (std::getline(is, line)) {
        if (line.empty()) continue;
        std::istringstream ls(line); bb_colentry_t r; size_t x;
        if (!(ls >> r.first)) throw genome::gnException("readBackboneColsFile: bad row");
        while (ls >> x) r.second.push_back(x);
        rows.push_back(r);
    }
}

// rows that continue one another in every sequence they hold (same sequences, same strands, abutting ends) become one
inline void mergeAdjacentSegments(std::vector<bb_seqentry_t> &rows)
{
    if (rows.empty()) return;
    const size_t N = rows[0].size();
    size_t ref = 0;
    auto key = [&](const bb_seqentry_t &r) { for (size_t g = 0; g < N; g++) if (r[g].first) return std::make_pair(g, (int64)std::llabs(r[g].first)); return std::make_pair(N, (int64)0); };
    std::stable_sort(rows.begin(), rows.end(), [&](const bb_seqentry_t &x, const bb_seqentry_t &y) { return key(x) < key(y); });
    (void)ref;
    std::vector<bb_seqentry_t> out;
    for (const bb_seqentry_t &r : rows) {
        bool merged = false;
        if (!out.empty()) {
            bb_seqentry_t &p = out.back();
            bool ok = true, any = false;
            // the first defined sequence gives the direction of travel: forward there, p comes before r
            for (size_t g = 0; g < N && ok; g++) {
                if ((p[g].first == 0) != (r[g].first == 0)) { ok = false; break; }
                if (!p[g].first) continue;
                any = true;
                if ((p[g].first < 0) != (r[g].first < 0)) { ok = false; break; }
                const int64 pl = std::llabs(p[g].first), pr = std::llabs(p[g].second), rl = std::llabs(r[g].first), rr = std::llabs(r[g].second);
                ok = p[g].first > 0 ? rl == pr + 1 : (rr + 1 == pl || rl == pr + 1);
            }
            if (ok && any) {
                for (size_t g = 0; g < N; g++) {
                    if (!p[g].first) continue;
                    const int64 lo = std::min<int64>(std::llabs(p[g].first), std::llabs(r[g].first)), hi = std::max<int64>(std::llabs(p[g].second), std::llabs(r[g].second));
                    const bool neg = p[g].first < 0;
                    p[g] = std::make_pair(neg ? -lo : lo, neg ? -hi : hi);
                }
                merged = true;
            }
        }
        if (!merged) out.push_back(r);
    }
    rows.swap(out);
}

// stretches of a sequence between (and before) its backbone rows, at least min_length long, become rows of their own
inline void addUniqueSegments(std::vector<bb_seqentry_t> &rows, size_t min_length = 20)
{
    if (rows.empty()) return;
    const size_t N = rows[0].size();
    std::vector<bb_seqentry_t> extra;
    for (size_t g = 0; g < N; g++) {
        std::vector<std::pair<int64, int64>> cov;
        for (const bb_seqentry_t &r : rows) if (r[g].first) cov.push_back(std::make_pair((int64)std::llabs(r[g].first), (int64)std::llabs(r[g].second)));
        std::sort(cov.begin(), cov.end());
        int64 next = 1;
        for (const auto &cv : cov) {
            if (cv.first - next >= (int64)min_length && cv.first > next) { bb_seqentry_t u(N, std::make_pair((int64)0, (int64)0)); u[g] = std::make_pair(next, cv.first - 1); extra.push_back(u); }
            next = std::max(next, cv.second + 1);
        }
    }
    rows.insert(rows.end(), extra.begin(), extra.end());
}

}  // namespace mems
#endif
