// mems_hip.h -- everything at once: the libMems-shaped C++ surface of the hot path, implemented on the C-ABI of
// libmauve_hip.so (include/mauve_hip.h) and nothing else.  The headers carry the names libMems gives them, so that
// in-tree sources include what they always included (src/UniqueMatchFinder.cpp and src/SeedMatchEnumerator.h compile
// unmodified against -I include; tests/test_compat_headers.py does exactly that).
#ifndef MEMS_HIP_H
#define MEMS_HIP_H
#include "../libGenome/gnSequence.h"
#include "AbstractMatch.h"
#include "Match.h"
#include "MatchProjectionAdapter.h"
#include "SortedMerList.h"
#include "DNAFileSML.h"
#include "MatchList.h"
#include "MatchFinder.h"
#include "MemHash.h"
#include "MaskedMemHash.h"
#include "PairwiseMatchFinder.h"
#include "RepeatHash.h"
#include "HipFinders.h"
#include "PairwiseScoringScheme.h"
#include "GappedAlignment.h"
#include "Interval.h"
#include "CompactGappedAlignment.h"
#include "IntervalList.h"
#include "GappedAligner.h"
#include "MuscleInterface.h"
#include "Aligner.h"
#include "ProgressiveAligner.h"
#include "Backbone.h"
#include "Islands.h"
#endif
