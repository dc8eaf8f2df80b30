// libMems/GappedAlignment.h -- a gapped alignment block as the in-tree code builds and reads it:
// `GappedAlignment ga(seq_count, aln_len)` (repeatoire.cpp:1238), SetAlignment(vector<string>) / SetAlignmentLength
// (:1264-1265), SetStart / SetLength per sequence (:1251-1258), GetAlignment(ga, seq_table) -> rows
// (stripGapColumns.cpp:54-55, unalign.cpp:57); the object GappedAligner::Align fills (MatchRecord.h:302-321).
#ifndef MAUVE_HIP_GAPPEDALIGNMENT_H
#define MAUVE_HIP_GAPPEDALIGNMENT_H

#include "AbstractMatch.h"

namespace mems {

class GappedAlignment : public AbstractMatch {
public:
    GappedAlignment() : aln_len_(0) {}
    GappedAlignment(uint seq_count, gnSeqI align_length) : aln_len_(align_length), start_(seq_count, NO_MATCH), len_(seq_count, 0), rows_(seq_count, std::string((size_t)align_length, '-')) {}
    virtual GappedAlignment *Copy() const { return new GappedAlignment(*this); }
    virtual uint SeqCount() const { return (uint)start_.size(); }
    virtual gnSeqI Length(uint seqI) const { return len_[seqI]; }
    virtual gnSeqI AlignmentLength() const { return aln_len_; }
    void SetAlignmentLength(gnSeqI n) { aln_len_ = n; for (std::string &r : rows_) r.resize((size_t)n, '-'); }
    virtual int64 Start(uint seqI) const { return start_[seqI]; }
    virtual void SetStart(uint seqI, int64 s) { start_[seqI] = s; }
    virtual void SetLength(gnSeqI len, uint seqI) { len_[seqI] = len; }
    // rows of equal length over A C G T and '-'; row i is sequence i read in the block's direction (a reverse
    // component is given as its reverse complement).  Lengths follow from the rows.
    void SetAlignment(const std::vector<std::string> &rows)
    {
        if (start_.size() < rows.size()) { start_.resize(rows.size(), NO_MATCH); len_.resize(rows.size(), 0); }
        rows_ = rows; rows_.resize(start_.size());
        aln_len_ = rows.empty() ? 0 : rows[0].size();
        for (size_t i = 0; i < rows_.size(); i++) {
            rows_[i].resize((size_t)aln_len_, '-');
            gnSeqI n = 0; for (char c : rows_[i]) n += c != '-';
            len_[i] = n;
        }
    }
    const std::vector<std::string> &GetAlignment() const { return rows_; }
    virtual void CropStart(gnSeqI cols) { crop_cols(0, cols); }
    virtual void CropEnd(gnSeqI cols) { crop_cols(aln_len_ - cols, aln_len_); }
    virtual void CropLeft(gnSeqI amount, uint seqI) { if (start_[seqI] > 0) CropStart(cols_for(seqI, amount, true)); else CropEnd(cols_for(seqI, amount, false)); }
    virtual void CropRight(gnSeqI amount, uint seqI) { if (start_[seqI] > 0) CropEnd(cols_for(seqI, amount, false)); else CropStart(cols_for(seqI, amount, true)); }
    virtual void Invert()
    {
        for (size_t i = 0; i < rows_.size(); i++) {
            std::string r(rows_[i].rbegin(), rows_[i].rend());
            for (char &c : r) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c == 'a' ? 't' : c == 'c' ? 'g' : c == 'g' ? 'c' : c == 't' ? 'a' : c;
            rows_[i] = r; start_[i] = -start_[i];
        }
    }
    virtual void GetColumn(gnSeqI col, std::vector<gnSeqI> &pos, std::vector<bool> &column) const
    {
        pos.assign(start_.size(), 0); column.assign(start_.size(), false);
        for (size_t i = 0; i < start_.size(); i++) {
            if (start_[i] == NO_MATCH || rows_[i][(size_t)col] == '-') continue;
            gnSeqI before = 0; for (gnSeqI k = 0; k < col; k++) before += rows_[i][(size_t)k] != '-';
            column[i] = true;
            pos[i] = start_[i] > 0 ? (gnSeqI)start_[i] + before : (gnSeqI)(-start_[i]) + len_[i] - 1 - before;
        }
    }
private:
    gnSeqI cols_for(uint seqI, gnSeqI residues, bool from_front) const       // columns that hold the first / last `residues` of a row
    {
        gnSeqI seen = 0, cols = 0;
        const std::string &r = rows_[seqI];
        for (gnSeqI k = 0; k < aln_len_ && seen < residues; k++) { const char c = from_front ? r[(size_t)k] : r[(size_t)(aln_len_ - 1 - k)]; seen += c != '-'; cols++; }
        return cols;
    }
    void crop_cols(gnSeqI a, gnSeqI b)                                       // drop columns [a, b)
    {
        for (size_t i = 0; i < rows_.size(); i++) {
            gnSeqI gone = 0; for (gnSeqI k = a; k < b; k++) gone += rows_[i][(size_t)k] != '-';
            if (start_[i] != NO_MATCH && gone) {
                const bool front = a == 0;
                if ((front && start_[i] > 0)) start_[i] += (int64)gone;
                else if (!front && start_[i] < 0) start_[i] -= (int64)gone;
                len_[i] -= gone;
                if (len_[i] == 0) start_[i] = NO_MATCH;
            }
            rows_[i].erase((size_t)a, (size_t)(b - a));
        }
        aln_len_ -= b - a;
    }
    gnSeqI aln_len_;
    std::vector<int64> start_;
    std::vector<gnSeqI> len_;
    std::vector<std::string> rows_;
};

// libMems' free function (stripGapColumns.cpp:54-55): the rows of a block as '-'-gapped strings
inline void GetAlignment(const GappedAlignment &ga, const std::vector<genome::gnSequence *> &, std::vector<std::string> &rows) { rows = ga.GetAlignment(); }

}  // namespace mems
#endif
