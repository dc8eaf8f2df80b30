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
template <class T> inline T absolut(T v) { return v < 0 ? -v : v; }   // toGrimmFormat.cpp:68, sortContigs.cpp:87, bbFilter.cpp:30

class gnSequence {
public:
    gnSequence() {}
    explicit gnSequence(const std::string &bases) : seq_(bases) { contig_start_.push_back(0); }
    gnSeqI length() const { return seq_.size(); }
    std::string ToString(gnSeqI len = 0, gnSeqI start = 1) const
    {
        if (start < 1 || start > seq_.size() + 1) throw gnException("gnSequence::ToString: start out of range");
        if (len == 0 || start - 1 + len > seq_.size()) len = seq_.size() - (start - 1);
        if (contig_start_.size() == 1 && !rc_.empty() && rc_[0]) {        // a single contig flagged reverse-complement reads the other strand
            std::string r(seq_.rbegin(), seq_.rend());
            for (char &ch : r) ch = ch == 'A' ? 'T' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch == 'T' ? 'A' : ch == 'a' ? 't' : ch == 'c' ? 'g' : ch == 'g' ? 'c' : ch == 't' ? 'a' : ch;
            return r.substr(start - 1, len);
        }
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
        std::string line; seq_.clear(); contig_start_.clear(); names_.clear(); rc_.clear();
        while (std::getline(in, line)) {
            if (!line.empty() && line[0] == '>') {                // a record starts here (records without bases leave no trace)
                if (contig_start_.empty() || (int64_t)seq_.size() != contig_start_.back()) contig_start_.push_back((int64_t)seq_.size());
                std::string nm = line.substr(1); while (!nm.empty() && (nm.back() == '\r' || nm.back() == ' ')) nm.pop_back();
                names_.resize(contig_start_.size()); names_.back() = nm;
                continue;
            }
            for (char ch : line) if (ch != '\r' && ch != ' ') seq_.push_back(ch);
        }
        if (contig_start_.empty()) contig_start_.push_back(0);
        if (contig_start_[0] != 0) { contig_start_.insert(contig_start_.begin(), 0); names_.insert(names_.begin(), std::string()); }
    }
    // ---- contigs (the records of a multi-FastA): mfa2xmfa.cpp:53-105, sortContigs.cpp:87-164, unalign.cpp:59-70,
    // backbone_global_to_local.cpp:46-47, multiToRawSequence.cpp:16-21.  Indices are 0-based, positions 1-based. ----
    uint32 contigListLength() const { return contigListSize(); }
    gnSeqI contigStart(uint32 i) const { return (gnSeqI)contig_start_.at(i) + 1; }
    gnSeqI contigLength(uint32 i) const { return (gnSeqI)((i + 1 < contig_start_.size() ? contig_start_[i + 1] : (int64_t)seq_.size()) - contig_start_.at(i)); }
    gnSequence contig(uint32 i) const
    {
        gnSequence c(seq_.substr((size_t)contig_start_.at(i), (size_t)contigLength(i)));
        c.names_.assign(1, contigName(i)); c.rc_.assign(1, isReverseComplement(i));
        return c;
    }
    std::string contigName(uint32 i) const { return i < names_.size() ? names_[i] : std::string(); }
    void setContigName(uint32 i, const std::string &n) { if (names_.size() <= i) names_.resize((size_t)i + 1); names_[i] = n; }
    uint32 contigIndexByBase(gnSeqI pos) const                         // the contig that holds base pos
    {
        if (pos < 1 || pos > seq_.size()) throw gnException("gnSequence::contigIndexByBase: position out of range");
        uint32 i = 0;
        while (i + 1 < contig_start_.size() && (gnSeqI)contig_start_[i + 1] < pos) i++;
        return i;
    }
    void globalToLocal(uint32 &contigI, gnSeqI &pos) const { contigI = contigIndexByBase(pos); pos -= (gnSeqI)contig_start_[contigI]; }
    void localToGlobal(uint32 contigI, gnSeqI &pos) const { pos += (gnSeqI)contig_start_.at(contigI); }
    // appending starts a new contig (sortContigs.cpp:113,142; mfa2xmfa.cpp:58; unalign.cpp)
    gnSequence &operator+=(const gnSequence &o)
    {
        for (uint32 i = 0; i < o.contigListSize(); i++) {
            if (!(seq_.empty() && contig_start_.size() == 1 && names_.empty())) contig_start_.push_back((int64_t)seq_.size());
            else if (contig_start_.empty()) contig_start_.push_back(0);
            const uint32 me = contigListSize() - 1;
            seq_ += o.seq_.substr((size_t)o.contig_start_[i], (size_t)o.contigLength(i));
            setContigName(me, o.contigName(i));
            if (o.isReverseComplement(i)) setReverseComplement(true, me);
        }
        return *this;
    }
    gnSequence &operator+=(const std::string &bases) { gnSequence t(bases); t.names_.assign(1, std::string()); return *this += t; }
    // a reverse-complement flag per contig: ToString of that contig then reads the other strand
    void setReverseComplement(bool rc, uint32 i = 0) { if (rc_.size() <= i) rc_.resize((size_t)i + 1, false); rc_[i] = rc; }
    bool isReverseComplement(uint32 i = 0) const { return i < rc_.size() && rc_[i]; }
    void ToString(std::string &out, gnSeqI len = 0, gnSeqI start = 1) const { out = ToString(len, start); }
    uint32 contigListSize() const { return (uint32)contig_start_.size(); }
    const std::vector<int64_t> &contigStarts() const { return contig_start_; }       // 0-based, ascending, first = 0
    void setContigStarts(const std::vector<int64_t> &s) { contig_start_ = s; if (contig_start_.empty() || contig_start_[0] != 0) contig_start_.insert(contig_start_.begin(), 0); }
    const std::string &str() const { return seq_; }
private:
    std::string seq_;
    std::vector<int64_t> contig_start_;
    std::vector<std::string> names_;                   // FastA deflines (without '>'), per contig
    std::vector<bool> rc_;
};

}  // namespace genome
#endif
