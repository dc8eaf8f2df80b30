// libMems/Match.h -- mems::Match: an ungapped match, one length for all components
// (SeedMatchEnumerator.h:75-76,83,119,132,140; row printer MatchRecord.h:350-355).
#ifndef MAUVE_HIP_MATCH_H
#define MAUVE_HIP_MATCH_H

#include <ostream>
#include "AbstractMatch.h"

namespace mems {

class Match : public AbstractMatch {
public:
    explicit Match(uint seq_count = 0) : len_(0), start_(seq_count, NO_MATCH) {}
    virtual Match *Copy() const { return new Match(*this); }                // SeedMatchEnumerator.h:119
    virtual uint SeqCount() const { return (uint)start_.size(); }
    virtual gnSeqI Length(uint seqI) const { return start_[seqI] == NO_MATCH ? 0 : (gnSeqI)len_; }
    gnSeqI Length() const { return (gnSeqI)len_; }
    virtual gnSeqI AlignmentLength() const { return (gnSeqI)len_; }
    void SetLength(gnSeqI len) { len_ = (int64)len; }                       // SeedMatchEnumerator.h:76
    virtual void SetLength(gnSeqI len, uint) { len_ = (int64)len; }         // length first (repeatoire.cpp:241)
    virtual int64 Start(uint seqI) const { return start_[seqI]; }
    virtual void SetStart(uint seqI, int64 s) { start_[seqI] = s; }         // :83
    // column-unit crops; on an ungapped match the per-sequence forms are the same thing (MatchRecord.h:263-276)
    virtual void CropStart(gnSeqI cols) { crop(cols, 0); }
    virtual void CropEnd(gnSeqI cols) { crop(0, cols); }
    virtual void CropLeft(gnSeqI amount, uint seqI) { if (start_[seqI] > 0) crop(amount, 0); else crop(0, amount); }
    virtual void CropRight(gnSeqI amount, uint seqI) { if (start_[seqI] > 0) crop(0, amount); else crop(amount, 0); }
    virtual void Invert() { for (int64 &s : start_) s = -s; }               // MatchRecord.h:283-284
    virtual void GetColumn(gnSeqI col, std::vector<gnSeqI> &pos, std::vector<bool> &column) const
    {
        pos.assign(start_.size(), 0); column.assign(start_.size(), false);
        for (size_t i = 0; i < start_.size(); i++) {
            if (start_[i] == NO_MATCH) continue;
            column[i] = true;
            pos[i] = start_[i] > 0 ? (gnSeqI)start_[i] + col : (gnSeqI)(-start_[i]) + (gnSeqI)len_ - 1 - col;
        }
    }
private:
    void crop(gnSeqI first, gnSeqI last)
    {
        for (int64 &s : start_) { if (s > 0) s += (int64)first; else if (s < 0) s -= (int64)last; }
        len_ -= (int64)(first + last);
    }
    int64 len_;
    std::vector<int64> start_;
};
inline std::ostream &operator<<(std::ostream &os, const Match &m)          // row shape of MatchRecord.h:350-355
{
    os << m.Length();
    for (uint i = 0; i < m.SeqCount(); i++) os << '\t' << m.Start(i);
    return os;
}
typedef Match UngappedLocalAlignment;

}  // namespace mems
#endif
