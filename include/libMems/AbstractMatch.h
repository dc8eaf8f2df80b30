// libMems/AbstractMatch.h -- the match interface of libMems as the in-tree code uses it:
// signed 1-based Start(seq) (0 = NO_MATCH: sortContigs.cpp:46, projectAndStrip.cpp:81), LeftEnd = |Start|,
// RightEnd = LeftEnd + Length - 1 (scoreAlignment.cpp:165-166), Orientation (repeatoire.cpp:194), per-sequence
// and column crops (MatchRecord.h:258-276), Invert (:283-284), Copy / Free ownership (SeedMatchEnumerator.h:119).
#ifndef MAUVE_HIP_ABSTRACTMATCH_H
#define MAUVE_HIP_ABSTRACTMATCH_H

#include "../libGenome/gnSequence.h"
#include "../mauve_hip.h"

namespace mems {

static const int64 NO_MATCH = 0;                 // sortContigs.cpp:46, SeedMatchEnumerator.h:132
typedef int score_t;                             // MatchRecord.h:186

class AbstractMatch {
public:
    enum orientation { forward, reverse, undefined };     // repeatoire.cpp:194, xmfa2maf.cpp:73; forward == 0 (SeedMatchEnumerator.h:93)
    virtual ~AbstractMatch() {}
    virtual AbstractMatch *Copy() const = 0;
    virtual void Free() { delete this; }
    virtual uint SeqCount() const = 0;
    virtual gnSeqI Length(uint seqI) const = 0;            // residues of sequence seqI (0 when absent)
    virtual gnSeqI AlignmentLength() const = 0;            // columns
    virtual int64 Start(uint seqI) const = 0;
    virtual void SetStart(uint seqI, int64 start) = 0;     // sets left end and orientation (repeatoire.cpp:1545)
    virtual void SetLength(gnSeqI len, uint seqI) = 0;     // length first (repeatoire.cpp:241,1258)
    virtual void CropStart(gnSeqI cols) = 0;               // column units
    virtual void CropEnd(gnSeqI cols) = 0;
    virtual void CropLeft(gnSeqI amount, uint seqI) = 0;   // units of sequence seqI (MatchRecord.h:263-264,276)
    virtual void CropRight(gnSeqI amount, uint seqI) = 0;
    virtual void Invert() = 0;
    // presence of a residue of every sequence in column col, and its 1-based position (coordinateTranslate.cpp:41)
    virtual void GetColumn(gnSeqI col, std::vector<gnSeqI> &pos, std::vector<bool> &column) const = 0;

    int64 operator[](uint seqI) const { return Start(seqI); }                          // SeedMatchEnumerator.h:132
    gnSeqI LeftEnd(uint seqI) const { return (gnSeqI)std::llabs(Start(seqI)); }        // scoreAlignment.cpp:165
    gnSeqI RightEnd(uint seqI) const { return Start(seqI) == NO_MATCH ? 0 : LeftEnd(seqI) + Length(seqI) - 1; }   // :166
    void SetLeftEnd(uint seqI, gnSeqI pos) { SetStart(seqI, Start(seqI) < 0 ? -(int64)pos : (int64)pos); }
    orientation Orientation(uint seqI) const { const int64 s = Start(seqI); return s == NO_MATCH ? undefined : (s < 0 ? reverse : forward); }
    void SetOrientation(uint seqI, orientation o) { const int64 a = std::llabs(Start(seqI)); SetStart(seqI, o == reverse ? -a : a); }
    uint Multiplicity() const { uint m = 0; for (uint i = 0; i < SeqCount(); i++) m += Start(i) != NO_MATCH; return m; }
    uint FirstStart() const { uint i = 0; while (i < SeqCount() && Start(i) == NO_MATCH) i++; return i; }
};

// orders matches by their start in one sequence (MatchRecord.h:292-293 MatchStartComparator< AbstractMatch > asc(0))
template <class MatchType>
class MatchStartComparator {
public:
    explicit MatchStartComparator(uint seq = 0) : seq_(seq) {}
    bool operator()(const MatchType *a, const MatchType *b) const
    {
        const int64 sa = std::llabs(a->Start(seq_)), sb = std::llabs(b->Start(seq_));
        return sa < sb;
    }
private:
    uint seq_;
};

// the same order over match objects held by value (unalign.cpp:53-54 sorts an IntervalList with it)
template <class MatchType>
class AbstractMatchStartComparator {
public:
    explicit AbstractMatchStartComparator(uint seq = 0) : seq_(seq) {}
    bool operator()(const MatchType &a, const MatchType &b) const { return std::llabs(a.Start(seq_)) < std::llabs(b.Start(seq_)); }
    bool operator()(const MatchType *a, const MatchType *b) const { return (*this)(*a, *b); }
private:
    uint seq_;
};

}  // namespace mems
#endif
