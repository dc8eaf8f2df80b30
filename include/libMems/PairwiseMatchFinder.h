// libMems/PairwiseMatchFinder.h -- MemHash on every pair of sequences separately (progressiveMauve.cpp:496-501:
// `PairwiseMatchFinder pmf; pmf.FindMatches(pairwise_match_list)` for up to four genomes): one sorted mer list,
// N(N-1)/2 joins on the device; every match has exactly two components.
#ifndef MAUVE_HIP_PAIRWISEMATCHFINDER_H
#define MAUVE_HIP_PAIRWISEMATCHFINDER_H
#include "MemHash.h"
namespace mems {
class PairwiseMatchFinder : public MemHash {
public:
    virtual PairwiseMatchFinder *Clone() const { return new PairwiseMatchFinder(*this); }
protected:
    virtual int kernelRule() const { return typeid(*this) == typeid(PairwiseMatchFinder) ? MAUVE_MODE_PAIRWISE : -1; }
};
}  // namespace mems
#endif
