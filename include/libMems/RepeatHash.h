// libMems/RepeatHash.h -- the repeat finder of libMems (base of the in-tree stub RepeatHashCat, RepeatHashCat.h:13;
// included by SeedMatchEnumerator.h:5): matches among the copies of a mer inside ONE sequence.  Every mer with two
// or more occurrences becomes one match of seed length whose components are the occurrences in position order,
// reverse-strand copies with negative starts (the rule SeedMatchEnumerator::HashMatch spells out,
// SeedMatchEnumerator.h:71-141); the device enumerates them (mauve_seed_match_enumerate).  Repeats are not extended.
// Concatenated multi-contig input (RepeatHashCat's concat_contig_start) is handled where the sequence is loaded:
// gnSequence keeps the contig starts and no seed window crosses one.
#ifndef MAUVE_HIP_REPEATHASH_H
#define MAUVE_HIP_REPEATHASH_H
#include "MemHash.h"
namespace mems {
class RepeatHash : public MemHash {
public:
    RepeatHash() : min_mult_(2), max_mult_(1000) {}
    virtual RepeatHash *Clone() const { return new RepeatHash(*this); }
    void SetMultiplicityRange(size_t lo, size_t hi) { min_mult_ = lo; max_mult_ = hi; }
    virtual boolean CreateMatches()
    {
        if (typeid(*this) != typeid(RepeatHash)) return MemHash::CreateMatches();      // a subclass: host callbacks
        if (seq_count != 1) return false;
        HipContext &hc = HipContext::global();
        const uint64_t pat = (uint64_t)sar_table[0]->Seed();
        const int seq = sar_table[0]->SequenceIndex() < 0 ? 0 : sar_table[0]->SequenceIndex();
        int64_t n = 0, ns = 0;
        hc.check(mauve_seed_match_enumerate(hc.get(), seq, pat, (int64_t)min_mult_, (int64_t)max_mult_, 0, &n, &ns, nullptr, nullptr, nullptr), "mauve_seed_match_enumerate");
        mult_.assign((size_t)n, 0); off_.assign((size_t)n + 1, 0); starts_.assign((size_t)ns, 0);
        hc.check(mauve_seed_match_enumerate(hc.get(), seq, pat, (int64_t)min_mult_, (int64_t)max_mult_, 0, &n, &ns, mult_.data(), off_.data(), starts_.data()), "mauve_seed_match_enumerate");
        return true;
    }
    virtual void GetMatchList(MatchList &ml) const
    {
        if (typeid(*this) != typeid(RepeatHash)) { MemHash::GetMatchList(ml); return; }
        for (size_t i = 0; i < mult_.size(); i++) {
            Match *m = new Match((uint)mult_[i]);
            m->SetLength(sar_table[0]->SeedLength());
            for (int64_t k = 0; k < mult_[i]; k++) m->SetStart((uint)k, starts_[(size_t)(off_[i] + k)]);
            ml.push_back(m);
        }
    }
private:
    size_t min_mult_, max_mult_;
    std::vector<int64_t> mult_, off_, starts_;
};
}  // namespace mems
#endif
