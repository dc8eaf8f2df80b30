// libMems/MatchFinder.h -- the seed / extend seam: mems::MatchFinder with the virtual hooks the in-tree
// subclasses override -- Clone (UniqueMatchFinder.h:28), EnumerateMatches(IdmerList&) (:31, SeedMatchEnumerator.h:38),
// HashMatch(IdmerList&) (:39), CreateMatches (:35), GetSar (:40) -- the members they touch (seq_count, sar_table:
// :56,60), FindMatchSeeds (:61) and the idmer record with its two comparators (UniqueMatchFinder.cpp:38,
// SeedMatchEnumerator.h:73).
//
// Two ways through a finder.  The classes of this library that name a rule the join kernel implements (MemHash,
// MaskedMemHash, PairwiseMatchFinder, HipUniqueMatchFinder) run entirely on the device: CreateMatches is one call
// of mauve_seed_mums.  Any OTHER subclass -- one that overrides EnumerateMatches or HashMatch, like the in-tree
// src/UniqueMatchFinder.cpp and src/SeedMatchEnumerator.h, which compile unmodified against these headers -- gets
// libMems' host callback path: FindMatchSeeds merges the sorted mer lists of the added sequences, calls
// EnumerateMatches once per mer shared by two or more windows, and MemHash::HashMatch collects the hits the
// subclass lets through, which the device then extends together (mauve_extend_hits).
#ifndef MAUVE_HIP_MATCHFINDER_H
#define MAUVE_HIP_MATCHFINDER_H

#include <list>
#include "MatchList.h"

namespace mems {

struct idmer {                                   // one occurrence of a mer (UniqueMatchFinder.cpp:46, SeedMatchEnumerator.h:83,133)
    gnSeqI position;                             // 0-based window start
    uint64 mer;                                  // SortedMerList::GetMer value: left-aligned mer | strand flag
    uint32 id;                                   // index of the sequence among the added ones
};
typedef std::list<idmer> IdmerList;
inline boolean idmer_id_lessthan(const idmer &a, const idmer &b) { return a.id < b.id; }                    // UniqueMatchFinder.cpp:38
inline boolean idmer_position_lessthan(const idmer &a, const idmer &b) { return a.position < b.position; }  // SeedMatchEnumerator.h:73

class MatchFinder {
public:
    MatchFinder() : seq_count(0), mask_(0), log_(nullptr) {}
    virtual ~MatchFinder() {}
    virtual MatchFinder *Clone() const = 0;                                // UniqueMatchFinder.h:28
    virtual boolean AddSequence(SortedMerList *sar, genome::gnSequence *seq)      // SeedMatchEnumerator.h:25
    {
        if (!sar || !seq) return false;
        if (!sar_table.empty() && sar->Seed() != sar_table[0]->Seed()) return false;
        sar_table.push_back(sar); seq_table.push_back(seq); seq_count++;
        return true;
    }
    void LogProgress(std::ostream *os) { log_ = os; }                      // mauveAligner.cpp:532
    void SetMask(uint64 m) { mask_ = m; }                                  // mauveAligner.cpp:530 (MaskedMemHash)
    void ClearSequences() { sar_table.clear(); seq_table.clear(); seq_count = 0; }
    virtual void Clear() { found_len_.clear(); found_start_.clear(); }
    // progressiveMauve.cpp:490-501: finder.FindMatches(match_list)
    virtual void FindMatches(MatchList &ml)
    {
        ClearSequences();
        for (size_t i = 0; i < ml.seq_table.size(); i++)
            if (!AddSequence(ml.sml_table.at(i), ml.seq_table[i])) { genome::ErrorMsg("Error adding " + (i < ml.seq_filename.size() ? ml.seq_filename[i] : std::string("sequence")) + "\n"); return; }
        CreateMatches();
        GetMatchList(ml);
    }
    virtual boolean CreateMatches() { if (seq_count < 1) return false; FindMatchSeeds(); return true; }   // SeedMatchEnumerator.h:35,59-65
    virtual void GetMatchList(MatchList &ml) const                         // progressiveMauve.cpp:545
    {
        for (size_t i = 0; i < found_len_.size(); i++) {
            Match *m = new Match(seq_count);
            m->SetLength((gnSeqI)found_len_[i]);
            for (uint g = 0; g < seq_count; g++) m->SetStart(g, found_start_[i * seq_count + g]);
            ml.push_back(m);
        }
    }
protected:
    virtual boolean EnumerateMatches(IdmerList &match_list) = 0;           // UniqueMatchFinder.h:31
    virtual boolean HashMatch(IdmerList &match_list) = 0;                  // SeedMatchEnumerator.h:39
    virtual SortedMerList *GetSar(uint32 sarI) const { return sar_table.at(sarI); }    // :40
    // The sorted-mer-list scan (SeedMatchEnumerator.h:61): a merge of the added sequences' lists; every mer that two
    // or more windows share goes to EnumerateMatches, occurrences in sequence order, positions ascending.
    void FindMatchSeeds()
    {
        const uint N = seq_count;
        std::vector<gnSeqI> at(N, 0), len(N, 0);
        for (uint g = 0; g < N; g++) len[g] = sar_table[g]->Length();
        const uint64 none = ~(uint64)0;
        for (;;) {
            uint64 lowest = none;
            for (uint g = 0; g < N; g++) if (at[g] < len[g]) lowest = std::min(lowest, sar_table[g]->SortedMer(at[g]) >> 1);
            if (lowest == none) break;
            IdmerList cur;
            for (uint g = 0; g < N; g++)
                while (at[g] < len[g] && (sar_table[g]->SortedMer(at[g]) >> 1) == lowest) {
                    idmer e; e.position = sar_table[g]->SortedPosition(at[g]); e.mer = sar_table[g]->SortedMer(at[g]); e.id = g;
                    cur.push_back(e); at[g]++;
                }
            if (cur.size() > 1 && !EnumerateMatches(cur)) break;
        }
    }
    uint seq_count;                              // SeedMatchEnumerator.h:60
    std::vector<SortedMerList *> sar_table;      // :56
    std::vector<genome::gnSequence *> seq_table;
    uint64 mask_;
    std::ostream *log_;
    std::vector<int64_t> found_len_, found_start_;
    uint found_seq_count_ = 0;                   // the sequence count the held matches were found with
};

}  // namespace mems
#endif
