// libMems/DNAFileSML.h -- the file-backed sorted mer list (uniqueMerCount.cpp:30-39 `DNAFileSML sml; sml.LoadFile(f)`).
// Here the in-memory list reads and writes its own cache file (SortedMerList::LoadFile / WriteFile), so the
// file-backed class is the same object.
#ifndef MAUVE_HIP_DNAFILESML_H
#define MAUVE_HIP_DNAFILESML_H
#include "SortedMerList.h"
namespace mems { typedef SortedMerList DNAFileSML; }
#endif
