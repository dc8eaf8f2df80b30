// libMems/MemHash.h -- mems::MemHash, the multi-MUM finder (mauveAligner.cpp:523-589 through MaskedMemHash;
// base of the in-tree UniqueMatchFinder, UniqueMatchFinder.h:21).  A plain MemHash is one device call; a subclass of
// the caller's own gets the host callback path of MatchFinder.h, with HashMatch collecting its hits for one
// extension launch on the device.
#ifndef MAUVE_HIP_MEMHASH_H
#define MAUVE_HIP_MEMHASH_H

#include <typeinfo>
#include "MatchFinder.h"

namespace mems {

class MemHash : public MatchFinder {
public:
    virtual MemHash *Clone() const { return new MemHash(*this); }
    virtual boolean CreateMatches()
    {
        if (seq_count < 1) return false;
        HipContext &hc = HipContext::global();
        const uint64_t pat = (uint64_t)sar_table[0]->Seed();
        int64_t n = 0;
        const int rule = kernelRule();
        if (rule >= 0)                                          // the finder's rule runs inside the join kernel
            hc.check(mauve_seed_mums(hc.get(), pat, rule, mask_, extendMatches() ? 1 : 0, &n), "mauve_seed_mums");
        else {                                                  // a subclass with a rule of its own: host callbacks, device extension
            hit_mask_.clear(); hit_pos_.clear(); hit_strand_.clear();
            FindMatchSeeds();
            hc.check(mauve_extend_hits(hc.get(), pat, (int64_t)hit_mask_.size(), hit_mask_.data(), hit_pos_.data(), hit_strand_.data(),
                                       extendMatches() ? 1 : 0, &n), "mauve_extend_hits");
        }
        // A finder keeps what it found until Clear(): a second search -- the next seed of a family, progressiveMauve.cpp:523-546
        // -- adds to it, and a match that an earlier one contains is not taken again (mauve_merge_matches).
        std::vector<int64_t> len((size_t)n, 0), st((size_t)n * seq_count, 0);
        hc.check(mauve_get_matches(hc.get(), len.data(), st.data()), "mauve_get_matches");
        if (found_len_.empty() || found_seq_count_ != seq_count) { found_len_.swap(len); found_start_.swap(st); }
        else {
            int64_t total = (int64_t)found_len_.size() + n;
            std::vector<int64_t> ml((size_t)total), ms((size_t)total * seq_count);
            hc.check(mauve_merge_matches((int)seq_count, (int64_t)found_len_.size(), found_len_.data(), found_start_.data(), n, len.data(), st.data(),
                                         &total, ml.data(), ms.data()), "mauve_merge_matches");
            ml.resize((size_t)total); ms.resize((size_t)total * seq_count);
            found_len_.swap(ml); found_start_.swap(ms);
        }
        found_seq_count_ = seq_count;
        if (log_) *log_ << "100%..done, " << n << " matches\n";
        return true;
    }
protected:
    // MAUVE_MODE_* when the dynamic type is one whose EnumerateMatches the join kernel implements, -1 otherwise.
    // Classes of this library override it with their rule; a foreign subclass inherits this answer: -1.
    virtual int kernelRule() const { return typeid(*this) == typeid(MemHash) ? MAUVE_MODE_MEM : -1; }
    virtual bool extendMatches() const { return true; }
    // MemHash's own rule on the host path (a subclass that only overrides HashMatch): a sequence holding the mer
    // more than once kills the seed
    virtual boolean EnumerateMatches(IdmerList &match_list)
    {
        match_list.sort(&idmer_id_lessthan);
        IdmerList::iterator a = match_list.begin(), b = a;
        for (++b; b != match_list.end(); ++a, ++b) if (a->id == b->id) return true;
        return match_list.size() >= 2 ? HashMatch(match_list) : true;
    }
    // one seed hit: the listed occurrences (one per sequence) become a match after extension.  SetMask keeps only the
    // hits whose component set equals the mask (MaskedMemHash, mauveAligner.cpp:525-531).
    virtual boolean HashMatch(IdmerList &match_list)
    {
        uint32_t m = 0;
        std::vector<int64_t> pos(seq_count, 0); std::vector<uint8_t> strand(seq_count, 0);
        for (const idmer &e : match_list) {
            if (e.id >= seq_count || (m >> e.id & 1)) return true;         // not a hit of this finder: ignored, as a repeat is
            m |= 1u << e.id; pos[e.id] = (int64_t)e.position; strand[e.id] = (uint8_t)(e.mer & 1);
        }
        if (__builtin_popcount(m) < 2 || (mask_ && m != (uint32_t)mask_)) return true;
        hit_mask_.push_back(m);
        hit_pos_.insert(hit_pos_.end(), pos.begin(), pos.end());
        hit_strand_.insert(hit_strand_.end(), strand.begin(), strand.end());
        return true;
    }
    std::vector<uint32_t> hit_mask_; std::vector<int64_t> hit_pos_; std::vector<uint8_t> hit_strand_;
};

}  // namespace mems
#endif
