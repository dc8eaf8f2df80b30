// libGenome/gnFASSource.h -- gnFASSource::Write as the in-tree tools call it (mfa2xmfa.cpp:61, sortContigs.cpp:175,
// unalign.cpp:73-77, toMultiFastA.cpp:49, getAlignmentWindows.cpp:111,132): one FastA record per contig, the contig
// name as its defline, bases wrapped at 80 columns.  The two trailing flags of the libGenome call (fill gaps, enforce
// unique names) have nothing to act on here and are accepted.
#ifndef MAUVE_HIP_GNFASSOURCE_H
#define MAUVE_HIP_GNFASSOURCE_H
#include "gnSequence.h"
namespace genome {
class gnFASSource {
public:
    static void Write(const gnSequence &seq, std::ostream &os, bool = true, bool = true)
    {
        for (uint32 i = 0; i < seq.contigListSize(); i++) {
            os << '>' << seq.contigName(i) << '\n';
            const std::string bases = seq.contig(i).ToString();
            for (size_t p = 0; p < bases.size(); p += 80) os << bases.substr(p, 80) << '\n';
        }
    }
    static void Write(const gnSequence &seq, const std::string &filename, bool a = true, bool b = true)
    {
        std::ofstream os(filename.c_str());
        if (!os) throw gnException("gnFASSource::Write: cannot open " + filename);
        Write(seq, os, a, b);
    }
};
}  // namespace genome
#endif
