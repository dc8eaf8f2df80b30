// libMems/MuscleInterface.h -- the name the in-tree call sites use for the installed gapped aligner
// (`MuscleInterface::getMuscleInterface()`, mauveAligner.cpp:82,674; MatchRecord.h:311; repeatoire.cpp:1262).
// libMUSCLE is not part of this library: the singleton behind the name is the batched HIP DP (GappedAligner.h).
#ifndef MAUVE_HIP_MUSCLEINTERFACE_H
#define MAUVE_HIP_MUSCLEINTERFACE_H
#include "GappedAligner.h"
namespace mems { typedef HipGappedAligner MuscleInterface; }
#endif
