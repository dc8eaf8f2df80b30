// libMems/Interval.h -- one block of the alignment, resident on the host: for every genome its range and strand
// (absent: left = right = 0) and, per alignment column, the set of genomes that have a base there (the bases
// themselves stay in seq_table).  Accessors as used in-tree: LeftEnd / RightEnd / Start / Orientation
// (toGrimmFormat.cpp:62-77), GetAlignment (repeatoire.cpp:1264-1265), SetMatches / StealMatches
// (MatchRecord.h:338-343, getAlignmentWindows.cpp:64,71), GetColumn (coordinateTranslate.cpp:41).
#ifndef MAUVE_HIP_INTERVAL_H
#define MAUVE_HIP_INTERVAL_H

#include "AbstractMatch.h"
#include "gnAlignedSequences.h"
#include <algorithm>

namespace mems {

// One Interval is one block of the alignment: for every genome its range and strand (absent: left = right = 0)
// and, per alignment column, the set of genomes that have a base there.  The bases themselves stay in seq_table.
class Interval : public AbstractMatch {
public:
    Interval() {}
    // Interval iv(begin, end) over AbstractMatch* (stripGapColumns.cpp:61-63): the matches are flattened into one block
    template <class It> Interval(It begin, It end) { std::vector<AbstractMatch *> v(begin, end); SetMatches(v); }
    Interval(const std::vector<int64> &left, const std::vector<int64> &right, const std::vector<char> &reverse,
             const std::vector<uint32_t> &cols) : left_(left), right_(right), rev_(reverse), cols_(cols) {}
    // An interval is a match like any other (libMems: Interval is an AbstractMatch -- extractBackbone.cpp:71-72 pushes
    // them into a vector<AbstractMatch*>, stripSubsetLCBs.cpp:130-137 crops them, projectAndStrip.cpp:104 inverts them);
    // LeftEnd / RightEnd / Orientation / Multiplicity / FirstStart come from the base (toGrimmFormat.cpp:62-77).
    virtual AbstractMatch *Copy() const { return new Interval(*this); }
    Interval *Clone() const { return new Interval(*this); }
    virtual uint SeqCount() const { return (uint)left_.size(); }
    virtual gnSeqI Length(uint seqI) const { return left_[seqI] ? (gnSeqI)(right_[seqI] - left_[seqI] + 1) : 0; }
    virtual int64 Start(uint seqI) const { return rev_[seqI] ? -left_[seqI] : left_[seqI]; }      // signed, NO_MATCH when absent
    virtual void SetStart(uint seqI, int64 start)
    {
        if (seqI >= left_.size()) { left_.resize(seqI + 1, 0); right_.resize(seqI + 1, 0); rev_.resize(seqI + 1, 0); }
        const int64 len = (int64)Length(seqI);
        left_[seqI] = std::llabs(start); rev_[seqI] = start < 0; right_[seqI] = start ? left_[seqI] + len - 1 : 0;
    }
    virtual void SetLength(gnSeqI len, uint seqI) { if (left_[seqI]) right_[seqI] = left_[seqI] + (int64)len - 1; }
    virtual gnSeqI AlignmentLength() const { return (gnSeqI)cols_.size(); }
    virtual void CropStart(gnSeqI n) { crop_cols(0, std::min<gnSeqI>(n, cols_.size())); }
    virtual void CropEnd(gnSeqI n) { const gnSeqI k = std::min<gnSeqI>(n, cols_.size()); crop_cols(cols_.size() - k, cols_.size()); }
    virtual void CropLeft(gnSeqI amount, uint seqI) { if (!rev_[seqI]) CropStart(cols_for(seqI, amount, true)); else CropEnd(cols_for(seqI, amount, false)); }
    virtual void CropRight(gnSeqI amount, uint seqI) { if (!rev_[seqI]) CropEnd(cols_for(seqI, amount, false)); else CropStart(cols_for(seqI, amount, true)); }
    virtual void Invert()
    {
        std::reverse(cols_.begin(), cols_.end());
        for (size_t g = 0; g < rev_.size(); g++) if (left_[g]) rev_[g] = !rev_[g];
        matches_.clear();
    }
    void CalculateOffset() {}                                  // extractSubalignments.cpp:24: the ranges are always current here
    const std::vector<uint32_t> &Columns() const { return cols_; }
    // SetMatches(vector&) STEALS the vector's contents (MatchRecord.h:338-339; getAlignmentWindows.cpp:64,71): the
    // matches, in order, become the block's columns -- the block keeps them (GetMatches / StealMatches give them back).
    void SetMatches(std::vector<AbstractMatch *> &matches)
    {
        for (AbstractMatch *m : matches_) m->Free();
        matches_.swap(matches); matches.clear();
        rebuild();
    }
    const std::vector<AbstractMatch *> &GetMatches() const { return matches_; }
    void StealMatches(std::vector<AbstractMatch *> &out) { out.swap(matches_); matches_.clear(); }
    // presence and 1-based position of every genome's residue in column col (coordinateTranslate.cpp:41)
    virtual void GetColumn(gnSeqI col, std::vector<gnSeqI> &pos, std::vector<bool> &column) const
    {
        const uint N = SeqCount();
        pos.assign(N, 0); column.assign(N, false);
        for (uint g = 0; g < N; g++) {
            if (!left_[g] || !(cols_[(size_t)col] >> g & 1)) continue;
            gnSeqI before = 0; for (gnSeqI k = 0; k < col; k++) before += cols_[(size_t)k] >> g & 1;
            column[g] = true;
            pos[g] = rev_[g] ? (gnSeqI)right_[g] - before : (gnSeqI)left_[g] + before;
        }
    }
    // rows of the block as '-'-gapped strings, one per genome (all gaps for an absent genome); a reverse
    // component is written as the reverse complement (GetAlignment, repeatoire.cpp:1264-1265)
    void GetAlignment(std::vector<std::string> &rows, const std::vector<genome::gnSequence *> &seq_table) const
    {
        const uint N = SeqCount();
        rows.assign(N, std::string(cols_.size(), '-'));
        for (uint g = 0; g < N; g++) {
            if (!left_[g]) continue;
            if (g >= seq_table.size() || (gnSeqI)right_[g] > seq_table[g]->length()) throw genome::gnException("Interval::GetAlignment: sequence table does not cover the interval");
            const std::string &sq = seq_table[g]->str();
            int64 nxt = rev_[g] ? right_[g] : left_[g];
            for (size_t k = 0; k < cols_.size(); k++) {
                if (!(cols_[k] >> g & 1)) continue;
                rows[g][k] = base_char(sq[(size_t)nxt - 1], rev_[g] != 0);
                nxt += rev_[g] ? -1 : 1;
            }
        }
    }
    // GetAlignedSequences(gnas, seq_table) (scoreAlignment.cpp:191, mauveAligner.cpp:770): the same rows; a sequence the
    // table does not cover (scoreAlignment hands in empty ones: it only asks where the gaps are) shows its bases as N
    void GetAlignedSequences(gnAlignedSequences &gnas, const std::vector<genome::gnSequence *> &seq_table) const
    {
        const uint N = SeqCount();
        gnas.sequences.assign(N, std::string(cols_.size(), '-')); gnas.names.assign(N, std::string());
        std::vector<std::string> rows;
        bool covered = true;
        for (uint g = 0; g < N && covered; g++) covered = !left_[g] || (g < seq_table.size() && seq_table[g] && (gnSeqI)right_[g] <= seq_table[g]->length());
        if (covered) { GetAlignment(rows, seq_table); gnas.sequences = rows; return; }
        for (uint g = 0; g < N; g++) if (left_[g]) for (size_t k = 0; k < cols_.size(); k++) if (cols_[k] >> g & 1) gnas.sequences[g][k] = 'N';
    }
    // upper case A C G T (complemented on the reverse strand); every other letter of the input is written as N, as
    // mauve_write_xmfa does (the device path keeps a bitmap of the ambiguous bases, not their letters)
    static char base_char(char c, bool complement)
    {
        int code;
        switch (c) { case 'A': case 'a': code = 0; break; case 'C': case 'c': code = 1; break; case 'G': case 'g': code = 2; break; case 'T': case 't': code = 3; break; default: return 'N'; }
        return "ACGT"[complement ? 3 - code : code];
    }
private:
    // columns [a, b) leave the block: every genome's range shrinks by the bases it had there, at the end they sat on
    void crop_cols(gnSeqI a, gnSeqI b)
    {
        if (b <= a) return;
        const bool at_start = a == 0;
        for (size_t g = 0; g < left_.size(); g++) {
            if (!left_[g]) continue;
            int64 k = 0; for (gnSeqI c = a; c < b; c++) k += cols_[(size_t)c] >> g & 1;
            if (at_start != (rev_[g] != 0)) left_[g] += k; else right_[g] -= k;
            if (right_[g] < left_[g]) { left_[g] = right_[g] = 0; rev_[g] = 0; }
        }
        cols_.erase(cols_.begin() + (std::ptrdiff_t)a, cols_.begin() + (std::ptrdiff_t)b);
        matches_.clear();                            // the flattened columns are the block now
    }
    // columns to drop so that `amount` bases of sequence seqI go: counted from the first column or from the last
    gnSeqI cols_for(uint seqI, gnSeqI amount, bool from_start) const
    {
        gnSeqI seen = 0, n = 0;
        const size_t L = cols_.size();
        while (n < L && seen < amount) { const size_t c = from_start ? n : L - 1 - n; seen += cols_[c] >> seqI & 1; n++; }
        return n;
    }
    void rebuild()                                   // ranges, strands and column masks from the matches, in order
    {
        left_.clear(); right_.clear(); rev_.clear(); cols_.clear();
        if (matches_.empty()) return;
        const uint N = matches_[0]->SeqCount();
        left_.assign(N, 0); right_.assign(N, 0); rev_.assign(N, 0);
        std::vector<gnSeqI> pos; std::vector<bool> col;
        for (const AbstractMatch *m : matches_) {
            for (uint g = 0; g < N && g < m->SeqCount(); g++) {
                if (m->Start(g) == NO_MATCH) continue;
                const int64 le = (int64)m->LeftEnd(g), re = (int64)m->RightEnd(g);
                if (!left_[g] || le < left_[g]) left_[g] = le;
                if (re > right_[g]) right_[g] = re;
                rev_[g] = m->Start(g) < 0;
            }
            for (gnSeqI k = 0; k < m->AlignmentLength(); k++) {
                m->GetColumn(k, pos, col);
                uint32_t mask = 0; for (uint g = 0; g < N && g < col.size(); g++) if (col[g]) mask |= 1u << g;
                cols_.push_back(mask);
            }
        }
    }
    std::vector<int64> left_, right_;
    std::vector<char> rev_;
    std::vector<uint32_t> cols_;
    std::vector<AbstractMatch *> matches_;           // only when built by SetMatches
};


// GetAlignment(iv, seq_table, rows) (stripGapColumns.cpp:36): the gapped rows of an interval, bases from the sequences
inline void GetAlignment(const Interval &iv, const std::vector<genome::gnSequence *> &seq_table, std::vector<std::string> &rows) { iv.GetAlignment(rows, seq_table); }

}  // namespace mems
#endif
