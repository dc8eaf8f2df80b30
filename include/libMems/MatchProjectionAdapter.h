// libMems/MatchProjectionAdapter.h -- a view of a subset / permutation of the components of a match
// (SeedMatchEnumerator.h:98 `MatchProjectionAdapter mpaa(mhe.Copy(), component_map)`; MatchRecord.h:242).
// The adapter owns the match it is given, as libMems' does (callers hand it a Copy()).
#ifndef MAUVE_HIP_MATCHPROJECTIONADAPTER_H
#define MAUVE_HIP_MATCHPROJECTIONADAPTER_H

#include "AbstractMatch.h"

namespace mems {

class MatchProjectionAdapter : public AbstractMatch {
public:
    MatchProjectionAdapter() : m(nullptr) {}
    MatchProjectionAdapter(AbstractMatch *match, const std::vector<size_t> &component_map) : m(match), seq_map(component_map) {}
    MatchProjectionAdapter(const MatchProjectionAdapter &o) : AbstractMatch(o), m(o.m ? o.m->Copy() : nullptr), seq_map(o.seq_map) {}
    MatchProjectionAdapter &operator=(const MatchProjectionAdapter &o)
    {
        if (this != &o) { if (m) m->Free(); m = o.m ? o.m->Copy() : nullptr; seq_map = o.seq_map; }
        return *this;
    }
    ~MatchProjectionAdapter() { if (m) m->Free(); }
    virtual MatchProjectionAdapter *Copy() const { return new MatchProjectionAdapter(*this); }
    virtual uint SeqCount() const { return (uint)seq_map.size(); }
    virtual gnSeqI Length(uint seqI) const { return m->Length((uint)seq_map[seqI]); }
    virtual gnSeqI AlignmentLength() const { return m->AlignmentLength(); }
    virtual int64 Start(uint seqI) const { return m->Start((uint)seq_map[seqI]); }
    virtual void SetStart(uint seqI, int64 s) { m->SetStart((uint)seq_map[seqI], s); }
    virtual void SetLength(gnSeqI len, uint seqI) { m->SetLength(len, (uint)seq_map[seqI]); }
    virtual void CropStart(gnSeqI cols) { m->CropStart(cols); }
    virtual void CropEnd(gnSeqI cols) { m->CropEnd(cols); }
    virtual void CropLeft(gnSeqI amount, uint seqI) { m->CropLeft(amount, (uint)seq_map[seqI]); }
    virtual void CropRight(gnSeqI amount, uint seqI) { m->CropRight(amount, (uint)seq_map[seqI]); }
    virtual void Invert() { m->Invert(); }
    virtual void GetColumn(gnSeqI col, std::vector<gnSeqI> &pos, std::vector<bool> &column) const
    {
        std::vector<gnSeqI> p; std::vector<bool> c;
        m->GetColumn(col, p, c);
        pos.resize(seq_map.size()); column.resize(seq_map.size());
        for (size_t i = 0; i < seq_map.size(); i++) { pos[i] = p[seq_map[i]]; column[i] = c[seq_map[i]]; }
    }
    AbstractMatch *m;
    std::vector<size_t> seq_map;
};

}  // namespace mems
#endif
