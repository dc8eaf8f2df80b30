// libGenome/gnRAWSource.h -- gnRAWSource::Write (toRawSequence.cpp:21, multiToRawSequence.cpp:21): the bases, nothing else.
#ifndef MAUVE_HIP_GNRAWSOURCE_H
#define MAUVE_HIP_GNRAWSOURCE_H
#include "gnSequence.h"
namespace genome {
class gnRAWSource {
public:
    static void Write(const gnSequence &seq, const std::string &filename)
    {
        std::ofstream os(filename.c_str(), std::ios::binary);
        if (!os) throw gnException("gnRAWSource::Write: cannot open " + filename);
        os << seq.ToString();
    }
};
}  // namespace genome
#endif
