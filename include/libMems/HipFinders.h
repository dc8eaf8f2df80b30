// libMems/HipFinders.h -- the two in-tree finders as device rules.  src/UniqueMatchFinder.cpp and
// src/SeedMatchEnumerator.h compile unmodified against these headers and then run libMems' host callback path
// (MatchFinder.h); a maintainer who wants the same results without the per-seed callbacks switches the class name:
//   mems::HipUniqueMatchFinder   UniqueMatchFinder::EnumerateMatches (UniqueMatchFinder.cpp:36-60) inside the join kernel
//   mems::HipSeedMatchEnumerator SeedMatchEnumerator::FindMatches / HashMatch / SetDirection (SeedMatchEnumerator.h:19-141)
//                                as one device enumeration of the runs of the sorted mer list
#ifndef MAUVE_HIP_HIPFINDERS_H
#define MAUVE_HIP_HIPFINDERS_H
#include "MemHash.h"
namespace mems {

class HipUniqueMatchFinder : public MemHash {
public:
    HipUniqueMatchFinder() {}
    HipUniqueMatchFinder(const HipUniqueMatchFinder &mh) : MemHash(mh) {}
    virtual HipUniqueMatchFinder *Clone() const { return new HipUniqueMatchFinder(*this); }
protected:
    virtual int kernelRule() const { return typeid(*this) == typeid(HipUniqueMatchFinder) ? MAUVE_MODE_UNIQUE : -1; }
};

class HipSeedMatchEnumerator : public MatchFinder {
public:
    virtual HipSeedMatchEnumerator *Clone() const { return new HipSeedMatchEnumerator(*this); }
    // SeedMatchEnumerator.h:19-33: single genome, every repeated seed becomes a Match of seed length
    void FindMatches(MatchList &match_list, size_t min_multi = 2, size_t max_multi = 1000, bool direct_repeats_only = false)
    {
        ClearSequences();
        for (size_t seqI = 0; seqI < match_list.seq_table.size(); ++seqI)
            if (!AddSequence(match_list.sml_table[seqI], match_list.seq_table[seqI])) {
                genome::ErrorMsg("Error adding " + (seqI < match_list.seq_filename.size() ? match_list.seq_filename[seqI] : std::string("sequence")) + "\n");
                return;
            }
        match_list.clear();
        if (seq_count != 1) return;                                         // CreateMatches, :59-65
        HipContext &hc = HipContext::global();
        int64_t n = 0, ns = 0;
        const uint64_t pat = (uint64_t)sar_table[0]->Seed();
        hc.check(mauve_seed_match_enumerate(hc.get(), 0, pat, (int64_t)min_multi, (int64_t)max_multi, direct_repeats_only, &n, &ns, nullptr, nullptr, nullptr), "mauve_seed_match_enumerate");
        std::vector<int64_t> mult((size_t)n), off((size_t)n + 1), st((size_t)ns);
        hc.check(mauve_seed_match_enumerate(hc.get(), 0, pat, (int64_t)min_multi, (int64_t)max_multi, direct_repeats_only, &n, &ns, mult.data(), off.data(), st.data()), "mauve_seed_match_enumerate");
        for (int64_t i = 0; i < n; i++) {
            Match *m = new Match((uint)mult[(size_t)i]);
            m->SetLength(GetSar(0)->SeedLength());
            for (int64_t k = 0; k < mult[(size_t)i]; k++) m->SetStart((uint)k, st[(size_t)(off[(size_t)i] + k)]);
            match_list.push_back(m);
        }
    }
protected:
    virtual boolean EnumerateMatches(IdmerList &) { return true; }
    virtual boolean HashMatch(IdmerList &) { return true; }
    virtual SortedMerList *GetSar(uint32) const { return sar_table[0]; }   // :54-57
};

}  // namespace mems
#endif
