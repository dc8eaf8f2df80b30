// libMems/gnAlignedSequences.h -- the rows of one aligned block as the in-tree programs take them from
// Interval::GetAlignedSequences(gnas, seq_table) (scoreAlignment.cpp:189-191,226-392; toMultiFastA.cpp:29-46;
// mauveAligner.cpp:769-779): `sequences` are equal-length '-'-gapped strings, one per sequence; output() writes them in
// one of the supported text formats (multi-FastA is the one this library writes).
#ifndef MAUVE_HIP_GNALIGNEDSEQUENCES_H
#define MAUVE_HIP_GNALIGNEDSEQUENCES_H
#include <iostream>
#include <string>
#include <vector>
#include "../libGenome/gnSequence.h"
namespace mems {
class gnAlignedSequences {
public:
    std::vector<std::string> sequences;
    std::vector<std::string> names;
    gnSeqI alignedSeqsSize() const { return sequences.empty() ? 0 : (gnSeqI)sequences[0].size(); }
    static const std::vector<std::string> &getSupportedFormats() { static const std::vector<std::string> f(1, "mfa"); return f; }
    static bool isSupportedFormat(const std::string &f) { return f == "mfa" || f == "fasta" || f == "multi-fasta"; }
    bool output(const std::string &format, std::ostream &os) const
    {
        if (!isSupportedFormat(format)) return false;
        for (size_t i = 0; i < sequences.size(); i++) {
            os << '>' << (i < names.size() && !names[i].empty() ? names[i] : "seq" + std::to_string(i + 1)) << '\n';
            for (size_t p = 0; p < sequences[i].size(); p += 80) os << sequences[i].substr(p, 80) << '\n';
        }
        return true;
    }
};
}  // namespace mems
#endif
