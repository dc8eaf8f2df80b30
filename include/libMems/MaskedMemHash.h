// libMems/MaskedMemHash.h -- MemHash restricted to one component set (mauveAligner.cpp:523-531:
// `MaskedMemHash match_finder; match_finder.SetMask((1 << seq_count) - 1)`): the mask goes into the join kernel.
#ifndef MAUVE_HIP_MASKEDMEMHASH_H
#define MAUVE_HIP_MASKEDMEMHASH_H
#include "MemHash.h"
namespace mems {
class MaskedMemHash : public MemHash {
public:
    virtual MaskedMemHash *Clone() const { return new MaskedMemHash(*this); }
protected:
    virtual int kernelRule() const { return typeid(*this) == typeid(MaskedMemHash) ? MAUVE_MODE_MEM : -1; }
};
}  // namespace mems
#endif
