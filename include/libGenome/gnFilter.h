// libGenome/gnFilter.h -- the two filters the in-tree sources ask for: DNAComplementFilter (repeatoire.cpp:1236-1279,
// scoreAlignment.cpp:142,394-403: Filter(char) complements a base, ReverseFilter(string&) reverse-complements in place)
// and fullDNASeqFilter (projectAndStrip.cpp: identity over IUPAC DNA letters).
#ifndef MAUVE_HIP_GNFILTER_H
#define MAUVE_HIP_GNFILTER_H
#include "gnSequence.h"
#include <algorithm>
namespace genome {
class gnFilter {
public:
    static const gnFilter *DNAComplementFilter() { static const gnFilter f(true); return &f; }
    static const gnFilter *fullDNASeqFilter() { static const gnFilter f(false); return &f; }
    char Filter(char ch) const
    {
        if (!complement_) return ch;
        switch (ch) {
            case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
            case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
            case 'R': return 'Y'; case 'Y': return 'R'; case 'K': return 'M'; case 'M': return 'K';
            case 'B': return 'V'; case 'V': return 'B'; case 'D': return 'H'; case 'H': return 'D';
            case 'r': return 'y'; case 'y': return 'r'; case 'k': return 'm'; case 'm': return 'k';
            case 'b': return 'v'; case 'v': return 'b'; case 'd': return 'h'; case 'h': return 'd';
            default: return ch;                                   // N, S, W, gaps: their own complement
        }
    }
    void Filter(std::string &s) const { for (char &ch : s) ch = Filter(ch); }
    void ReverseFilter(std::string &s) const { std::reverse(s.begin(), s.end()); Filter(s); }
private:
    explicit gnFilter(bool complement) : complement_(complement) {}
    bool complement_;
};
}  // namespace genome
#endif
