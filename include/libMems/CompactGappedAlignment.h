// libMems/CompactGappedAlignment.h -- the bit-matrix form of a gapped block (repeatoire.cpp:1316-1318,1347;
// MatchRecord.h:341-343; scoreProcrastAlignment.cpp:292-298; bbBreakOnGenes.cpp:154-155): one bitset per sequence,
// bit = residue present in the column; SeqPosToColumn, copyRange.  `bitset_t` (dynamic_bitset.h) stands in for boost::dynamic_bitset<>.
#ifndef MAUVE_HIP_COMPACTGAPPEDALIGNMENT_H
#define MAUVE_HIP_COMPACTGAPPEDALIGNMENT_H

#include <algorithm>
#include "GappedAlignment.h"
#include "Interval.h"
#include "dynamic_bitset.h"

namespace mems {


template <class BaseType = AbstractMatch>
class CompactGappedAlignment : public AbstractMatch {
public:
    CompactGappedAlignment() : aln_len_(0) {}
    CompactGappedAlignment(uint seq_count, gnSeqI align_length) : aln_len_(align_length), start_(seq_count, NO_MATCH), len_(seq_count, 0), bits_(seq_count, bitset_t((size_t)align_length, false)) {}
    // from any match: its GetColumn tells, column by column, who has a residue (MatchRecord.h:341 `CompactGappedAlignment<> tmpcga(*this)`)
    explicit CompactGappedAlignment(const AbstractMatch &m) { assign(m); }
    explicit CompactGappedAlignment(const Interval &iv) : aln_len_(iv.AlignmentLength())
    {
        const uint N = iv.SeqCount();
        start_.resize(N); len_.resize(N); bits_.assign(N, bitset_t((size_t)aln_len_, false));
        for (uint g = 0; g < N; g++) {
            start_[g] = iv.Start(g); len_[g] = iv.Length(g);
            const std::vector<uint32_t> &c = iv.Columns();
            for (size_t k = 0; k < c.size(); k++) bits_[g][k] = (c[k] >> g & 1) != 0;
        }
    }
    virtual CompactGappedAlignment *Copy() const { return new CompactGappedAlignment(*this); }
    virtual uint SeqCount() const { return (uint)start_.size(); }
    virtual gnSeqI Length(uint seqI) const { return len_[seqI]; }
    virtual gnSeqI AlignmentLength() const { return aln_len_; }
    virtual int64 Start(uint seqI) const { return start_[seqI]; }
    virtual void SetStart(uint seqI, int64 s) { start_[seqI] = s; }
    virtual void SetLength(gnSeqI len, uint seqI) { len_[seqI] = len; }
    const std::vector<bitset_t> &GetAlignment() const { return bits_; }                    // scoreProcrastAlignment.cpp:292
    void SetAlignment(const std::vector<bitset_t> &b) { bits_ = b; aln_len_ = b.empty() ? 0 : b[0].size(); for (size_t i = 0; i < b.size() && i < len_.size(); i++) { gnSeqI n = 0; for (bool x : b[i]) n += x; len_[i] = n; } }
    // column that holds base `pos` (1-based sequence coordinate) of sequence seqI (repeatoire.cpp:1347)
    gnSeqI SeqPosToColumn(uint seqI, gnSeqI pos) const
    {
        if (start_[seqI] == NO_MATCH || pos < LeftEnd(seqI) || pos > RightEnd(seqI)) throw genome::gnException("SeqPosToColumn: position outside the alignment");
        const gnSeqI want = start_[seqI] > 0 ? pos - LeftEnd(seqI) : RightEnd(seqI) - pos;   // residues before it in column order
        gnSeqI seen = 0;
        for (gnSeqI k = 0; k < aln_len_; k++) if (bits_[seqI][(size_t)k]) { if (seen == want) return k; seen++; }
        throw genome::gnException("SeqPosToColumn: inconsistent alignment");
    }
    // dest = columns [left_col, left_col + len) of this alignment (bbBreakOnGenes.cpp:154-155)
    void copyRange(CompactGappedAlignment &dest, gnSeqI left_col, gnSeqI len) const
    {
        dest = *this;
        dest.CropEnd(aln_len_ - left_col - len);
        dest.CropStart(left_col);
    }
    virtual void CropStart(gnSeqI cols) { crop_cols(0, cols); }
    virtual void CropEnd(gnSeqI cols) { crop_cols(aln_len_ - cols, aln_len_); }
    virtual void CropLeft(gnSeqI amount, uint seqI) { if (start_[seqI] > 0) CropStart(cols_for(seqI, amount, true)); else CropEnd(cols_for(seqI, amount, false)); }
    virtual void CropRight(gnSeqI amount, uint seqI) { if (start_[seqI] > 0) CropEnd(cols_for(seqI, amount, false)); else CropStart(cols_for(seqI, amount, true)); }
    virtual void Invert() { for (size_t i = 0; i < bits_.size(); i++) { bits_[i].reverse(); start_[i] = -start_[i]; } }
    virtual void GetColumn(gnSeqI col, std::vector<gnSeqI> &pos, std::vector<bool> &column) const
    {
        pos.assign(start_.size(), 0); column.assign(start_.size(), false);
        for (size_t i = 0; i < start_.size(); i++) {
            if (start_[i] == NO_MATCH || !bits_[i][(size_t)col]) continue;
            gnSeqI before = 0; for (gnSeqI k = 0; k < col; k++) before += bits_[i][(size_t)k];
            column[i] = true;
            pos[i] = start_[i] > 0 ? (gnSeqI)start_[i] + before : (gnSeqI)(-start_[i]) + len_[i] - 1 - before;
        }
    }
private:
    void assign(const AbstractMatch &m)
    {
        const uint N = m.SeqCount();
        aln_len_ = m.AlignmentLength();
        start_.resize(N); len_.resize(N); bits_.assign(N, bitset_t((size_t)aln_len_, false));
        for (uint g = 0; g < N; g++) { start_[g] = m.Start(g); len_[g] = m.Length(g); }
        std::vector<gnSeqI> pos; std::vector<bool> col;
        for (gnSeqI k = 0; k < aln_len_; k++) { m.GetColumn(k, pos, col); for (uint g = 0; g < N; g++) bits_[g][(size_t)k] = col[g]; }
    }
    gnSeqI cols_for(uint seqI, gnSeqI residues, bool from_front) const
    {
        gnSeqI seen = 0, cols = 0;
        for (gnSeqI k = 0; k < aln_len_ && seen < residues; k++) { seen += bits_[seqI][(size_t)(from_front ? k : aln_len_ - 1 - k)]; cols++; }
        return cols;
    }
    void crop_cols(gnSeqI a, gnSeqI b)
    {
        for (size_t i = 0; i < bits_.size(); i++) {
            gnSeqI gone = 0; for (gnSeqI k = a; k < b; k++) gone += bits_[i][(size_t)k];
            if (start_[i] != NO_MATCH && gone) {
                const bool front = a == 0;
                if (front && start_[i] > 0) start_[i] += (int64)gone;
                else if (!front && start_[i] < 0) start_[i] -= (int64)gone;
                len_[i] -= gone;
                if (len_[i] == 0) start_[i] = NO_MATCH;
            }
            bits_[i].erase_range((size_t)a, (size_t)b);
        }
        aln_len_ -= b - a;
    }
    gnSeqI aln_len_;
    std::vector<int64> start_;
    std::vector<gnSeqI> len_;
    std::vector<bitset_t> bits_;
};

}  // namespace mems
#endif
