// libMems/GappedAligner.h -- the gapped-DP seam: the interface Aligner::SetGappedAligner installs
// (mauveAligner.cpp:674) and GappedMatchRecord::finalize calls per inter-anchor interval
// (`Align(GappedAlignment& cr, AbstractMatch* left, AbstractMatch* right, seq_table)`, MatchRecord.h:311), plus
// CallMuscleFast (repeatoire.cpp:1262,1288).  HipGappedAligner is the built-in implementation: the batched DP kernel
// of libmauve_hip.so.  Any other GappedAligner installed on an Aligner is called interval by interval.
#ifndef MAUVE_HIP_GAPPEDALIGNER_H
#define MAUVE_HIP_GAPPEDALIGNER_H

#include "GappedAlignment.h"
#include "PairwiseScoringScheme.h"
#include "GuideTree.h"
#include <fstream>
#include "SortedMerList.h"

namespace mems {

// the sequence of genome g between two matches, in the direction of the left match's component (reverse components
// are read as the reverse complement); empty when either match lacks the genome or they abut
inline bool getInterveningSequence(const AbstractMatch *left, const AbstractMatch *right, uint g, const std::vector<genome::gnSequence *> &seq_table,
                                   std::string &out, int64 &start_out)
{
    out.clear(); start_out = NO_MATCH;
    if (left->Start(g) == NO_MATCH || right->Start(g) == NO_MATCH) return false;
    const bool rev = left->Start(g) < 0;
    if ((right->Start(g) < 0) != rev) return false;
    const int64 lo = rev ? (int64)right->RightEnd(g) + 1 : (int64)left->RightEnd(g) + 1;
    const int64 hi = rev ? (int64)left->LeftEnd(g) - 1 : (int64)right->LeftEnd(g) - 1;
    if (hi < lo) return true;
    if (g >= seq_table.size() || (gnSeqI)hi > seq_table[g]->length()) throw genome::gnException("GappedAligner::Align: sequence table does not cover the interval");
    out = seq_table[g]->ToString((gnSeqI)(hi - lo + 1), (gnSeqI)lo);
    if (rev) {
        std::string r(out.rbegin(), out.rend());
        for (char &c : r) c = c == 'A' || c == 'a' ? 'T' : c == 'C' || c == 'c' ? 'G' : c == 'G' || c == 'g' ? 'C' : c == 'T' || c == 't' ? 'A' : 'N';
        out = r;
    }
    start_out = rev ? -lo : lo;
    return true;
}

class GappedAligner {
public:
    GappedAligner() : max_alignment_length(10000) {}
    virtual ~GappedAligner() {}
    // CallMuscleFast shape (repeatoire.cpp:1262): aligned rows out, raw sequences in
    virtual bool CallMuscleFast(std::vector<std::string> &aln_out, const std::vector<std::string> &seqs_in, int gap_open, int gap_extend) = 0;
    // MatchRecord.h:311: align what lies between two anchors; cr receives rows, starts and lengths.  The default
    // implementation cuts the sequences out and hands them to CallMuscleFast.
    virtual boolean Align(GappedAlignment &cr, AbstractMatch *r_begin, AbstractMatch *r_end, std::vector<genome::gnSequence *> &seq_table)
    {
        const uint N = r_begin->SeqCount();
        std::vector<std::string> in(N), out;
        std::vector<int64> st(N, NO_MATCH);
        gnSeqI longest = 0; uint nonempty = 0;
        for (uint g = 0; g < N; g++) { getInterveningSequence(r_begin, r_end, g, seq_table, in[g], st[g]); longest = std::max<gnSeqI>(longest, in[g].size()); nonempty += !in[g].empty(); }
        if (nonempty < 2 || longest > max_alignment_length) return false;
        PairwiseScoringScheme pss;
        if (!CallMuscleFast(out, in, pss.gap_open, pss.gap_extend)) return false;
        cr = GappedAlignment(N, out.empty() ? 0 : out[0].size());
        cr.SetAlignment(out);
        for (uint g = 0; g < N; g++) cr.SetStart(g, in[g].empty() ? NO_MATCH : st[g]);
        return true;
    }
    void SetMaxAlignmentLength(gnSeqI n) { max_alignment_length = n; }
protected:
    gnSeqI max_alignment_length;
};

class HipGappedAligner : public GappedAligner {  // stands where MuscleInterface::getMuscleInterface() stood (mauveAligner.cpp:82)
public:
    static HipGappedAligner &getInterface() { static HipGappedAligner g; return g; }
    static HipGappedAligner &getMuscleInterface() { return getInterface(); }
    void SetScoring(const PairwiseScoringScheme &p) { pss_ = p; }
    void SetExtraMuscleArguments(const std::string &) {}                     // mauveAligner.cpp:374-376: nothing to pass on
    void ParseMusclePath(const char *) {}
    // mi.CreateTree(distance, tree_filename) (mauveAligner.cpp:619-622): UPGMA with the merge rule of mauve_guide_tree
    // (closest pair, ties to the lowest ids, average linkage) over the caller's matrix, written as NEWICK (GuideTree.h)
    template <class MatrixT> void CreateTree(const MatrixT &distance, const std::string &tree_filename)
    {
        const int N = (int)distance.rows(), M = 2 * N - 1;
        if (N < 2) throw genome::gnException("CreateTree: at least two sequences required");
        std::vector<double> D((size_t)M * M, 0.0); std::vector<int64_t> size((size_t)M, 1), ppm((size_t)N * N, 0);
        std::vector<int32_t> left((size_t)M, -1), right((size_t)M, -1); std::vector<char> active((size_t)M, 0);
        for (int i = 0; i < N; i++) { active[(size_t)i] = 1; for (int j = 0; j < N; j++) { D[(size_t)i * M + j] = distance(i, j); ppm[(size_t)i * N + j] = (int64_t)(distance(i, j) * 1e6 + 0.5); } }
        for (int k = N; k < M; k++) {
            int ba = -1, bb = -1; double bd = 0;
            for (int a = 0; a < k; a++) if (active[(size_t)a]) for (int b = a + 1; b < k; b++) if (active[(size_t)b])
                if (ba < 0 || D[(size_t)a * M + b] < bd) { ba = a; bb = b; bd = D[(size_t)a * M + b]; }
            left[(size_t)k] = ba; right[(size_t)k] = bb; size[(size_t)k] = size[(size_t)ba] + size[(size_t)bb];
            for (int x = 0; x < k; x++) if (active[(size_t)x] && x != ba && x != bb)
                D[(size_t)k * M + x] = D[(size_t)x * M + k] = ((double)size[(size_t)ba] * D[(size_t)ba * M + x] + (double)size[(size_t)bb] * D[(size_t)bb * M + x]) / (double)size[(size_t)k];
            active[(size_t)ba] = active[(size_t)bb] = 0; active[(size_t)k] = 1;
        }
        std::ofstream out(tree_filename.c_str());
        if (!out) throw genome::gnException("CreateTree: cannot write " + tree_filename);
        out << guideTreeToNewick(N, left, right, ppm);
    }
    virtual bool CallMuscleFast(std::vector<std::string> &aln_out, const std::vector<std::string> &seqs_in, int gap_open, int gap_extend)
    {
        const int N = (int)seqs_in.size();
        if (N < 1 || N > MAUVE_MAX_SEQ) return false;
        std::vector<uint8_t> codes; std::vector<int64_t> off(1, 0);
        for (const std::string &s : seqs_in) {
            for (char ch : s) codes.push_back(ch == 'C' || ch == 'c' ? 1 : ch == 'G' || ch == 'g' ? 2 : ch == 'T' || ch == 't' ? 3 : 0);
            off.push_back((int64_t)codes.size());
        }
        mauve_scoring sc;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) sc.matrix[i][j] = pss_.matrix[i][j];
        sc.gap_open = gap_open; sc.gap_extend = gap_extend;
        std::vector<uint32_t> cols(codes.size() + 1); int64_t col_off[2] = {0, 0}, score = 0;
        HipContext &hc = HipContext::global();
        if (mauve_dp_batch(hc.get(), N, 1, codes.empty() ? nullptr : codes.data(), off.data(), &sc, cols.data(), col_off, &score) != MAUVE_OK) return false;
        aln_out.assign((size_t)N, std::string());
        std::vector<size_t> nxt((size_t)N, 0);
        for (int64_t c = 0; c < col_off[1]; c++)
            for (int g = 0; g < N; g++) aln_out[(size_t)g].push_back((cols[(size_t)c] >> g & 1) ? seqs_in[(size_t)g][nxt[(size_t)g]++] : '-');
        return true;
    }
private:
    PairwiseScoringScheme pss_;
};

}  // namespace mems
#endif
