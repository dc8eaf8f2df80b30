// libGenome/gnSequence.h -- the slice of libGenome the hot path touches, on top of nothing but the standard library:
// the integer typedefs the in-tree sources use unqualified, genome::gnException / ErrorMsg, and gnSequence with
// length(), ToString(len, start) (1-based) and LoadSource (FastA; records concatenated, contig starts kept).
// Call sites that pin the shapes: mauveAligner.cpp:453-465,498-503,852-864; SeedMatchEnumerator.h:26;
// RepeatHashCat.h:19-20 (contig-start table of a concatenated multi-contig sequence).
#ifndef MAUVE_HIP_GNSEQUENCE_H
#define MAUVE_HIP_GNSEQUENCE_H

#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <list>
#include <stdexcept>
#include <string>
#include <vector>

typedef bool boolean;
typedef unsigned int uint;
typedef uint32_t uint32;
typedef int64_t int64;
typedef uint64_t uint64;
typedef uint64_t gnSeqI;

// libGenome's headers put the standard containers in scope for everything that includes them; in-tree code relies
// on it (SeedMatchEnumerator.h:88 writes `vector< size_t >` unqualified, UniqueMatchFinder.cpp `list`).
using std::vector;
using std::list;
using std::string;

namespace genome {

// gnException stand-in: thrown for I/O and argument errors, printable (mauveAligner.cpp:498-503,852-864)
class gnException : public std::runtime_error {
public:
    explicit gnException(const std::string &m) : std::runtime_error(m) {}
};
inline std::ostream &operator<<(std::ostream &os, const gnException &e) { return os << e.what(); }
inline void ErrorMsg(const std::string &m) { std::cerr << m; }        // SeedMatchEnumerator.h:26
inline void breakHere() {}                                            // repeatoire.cpp:178 (debug trap)

class gnSequence {
public:
    gnSequence() {}
    explicit gnSequence(const std::string &bases) : seq_(bases) { contig_start_.push_back(0); }
    gnSeqI length() const { return seq_.size(); }
    std::string ToString(gnSeqI len = 0, gnSeqI start = 1) const
    {
        if (start < 1 || start > seq_.size() + 1) throw gnException("gnSequence::ToString: start out of range");
        if (len == 0 || start - 1 + len > seq_.size()) len = seq_.size() - (start - 1);
        return seq_.substr(start - 1, len);
    }
    // FastA only (the formats the tree feeds this path: mauveAligner.cpp:453-465).  The records of a multi-record
    // file are concatenated, as LoadSequences does for a genome with several contigs, and the 0-based start of
    // every record is remembered: seeds, matches and gapped alignments never run across a contig join
    // (RepeatHashCat.h:19-20 concat_contig_start; MatchList::upload hands the table to the device).
    void LoadSource(const std::string &path)
    {
        std::ifstream in(path.c_str());
        if (!in) throw gnException("gnSequence::LoadSource: cannot open " + path);
        std::string line; seq_.clear(); contig_start_.clear();
        while (std::getline(in, line)) {
            if (!line.empty() && line[0] == '>') {                // a record starts here (records without bases leave no trace)
                if (contig_start_.empty() || (int64_t)seq_.size() != contig_start_.back()) contig_start_.push_back((int64_t)seq_.size());
                continue;
            }
            for (char ch : line) if (ch != '\r' && ch != ' ') seq_.push_back(ch);
        }
        if (contig_start_.empty()) contig_start_.push_back(0);
        if (contig_start_[0] != 0) contig_start_.insert(contig_start_.begin(), 0);
    }
    uint32 contigListSize() const { return (uint32)contig_start_.size(); }
    const std::vector<int64_t> &contigStarts() const { return contig_start_; }       // 0-based, ascending, first = 0
    void setContigStarts(const std::vector<int64_t> &s) { contig_start_ = s; if (contig_start_.empty() || contig_start_[0] != 0) contig_start_.insert(contig_start_.begin(), 0); }
    const std::string &str() const { return seq_; }
private:
    std::string seq_;
    std::vector<int64_t> contig_start_;
};

}  // namespace genome
#endif
