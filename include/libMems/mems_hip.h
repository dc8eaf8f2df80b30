// mems_hip.h -- the libMems-shaped C++ surface of the hot path, implemented on the C-ABI of
// libmauve_hip.so (include/mauve_hip.h) and nothing else.
//
// The reference's seams for this path are C++ virtual interfaces of libMems; this header
// mirrors the part of that surface the in-tree code uses, with the same names, argument order and error
// behaviour, so that call sites shaped like mauveAligner.cpp:523-589,668-698 and
// progressiveMauve.cpp:490-501 compile against it.  Citations give the in-tree line that pins each member.
//
// Differences that follow from the device design (documented, not hidden):
//  * EnumerateMatches(IdmerList&) / HashMatch(IdmerList&) are per-seed host callbacks in libMems; here the rule
//    they implement runs inside the join kernel, selected by the finder class (MemHash / UniqueMatchFinder /
//    SeedMatchEnumerator).  Subclasses choose a rule through seedRule() instead of overriding the callback.
//  * Match objects are plain values owned by the MatchList (Copy()/Free() keep their meaning).
#ifndef MEMS_HIP_H
#define MEMS_HIP_H

#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <ostream>
#include <sstream>
#include <algorithm>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../mauve_hip.h"

typedef bool boolean;
typedef unsigned int uint;
typedef uint32_t uint32;
typedef int64_t int64;
typedef uint64_t uint64;
typedef uint64_t gnSeqI;

namespace genome {

// gnException stand-in: thrown for I/O and argument errors, printable (mauveAligner.cpp:498-503,852-864)
class gnException : public std::runtime_error {
public:
    explicit gnException(const std::string &m) : std::runtime_error(m) {}
};
inline std::ostream &operator<<(std::ostream &os, const gnException &e) { return os << e.what(); }
inline void ErrorMsg(const std::string &m) { std::cerr << m; }        // SeedMatchEnumerator.h:26

// The slice of gnSequence the hot path touches: length(), ToString(len, start) (1-based), LoadSource.
class gnSequence {
public:
    gnSequence() {}
    explicit gnSequence(const std::string &bases) : seq_(bases) {}
    gnSeqI length() const { return seq_.size(); }
    std::string ToString(gnSeqI len = 0, gnSeqI start = 1) const
    {
        if (start < 1 || start > seq_.size() + 1) throw gnException("gnSequence::ToString: start out of range");
        if (len == 0 || start - 1 + len > seq_.size()) len = seq_.size() - (start - 1);
        return seq_.substr(start - 1, len);
    }
    // FastA only (the formats the tree feeds this path: mauveAligner.cpp:453-465); multi-record files are
    // concatenated, as LoadSequences does for a genome with several contigs.
    void LoadSource(const std::string &path)
    {
        std::ifstream in(path.c_str());
        if (!in) throw gnException("gnSequence::LoadSource: cannot open " + path);
        std::string line; seq_.clear();
        while (std::getline(in, line)) {
            if (!line.empty() && line[0] == '>') continue;
            for (char ch : line) if (ch != '\r' && ch != ' ') seq_.push_back(ch);
        }
    }
    const std::string &str() const { return seq_; }
private:
    std::string seq_;
};

}  // namespace genome

namespace mems {

static const int64 NO_MATCH = 0;                 // sortContigs.cpp:46, SeedMatchEnumerator.h:132
static const int CODING_SEED = MAUVE_CODING_SEED;
static const int SOLID_SEED = MAUVE_SOLID_SEED;  // repeatoire.cpp:1847

inline int64 getSeed(int weight, int rank = 0) { return (int64)mauve_get_seed(weight, rank); }       // progressiveMauve.cpp:217
inline uint getSeedLength(int64 seed) { return (uint)mauve_seed_length((uint64_t)seed); }            // :515-517
inline uint getDefaultSeedWeight(gnSeqI avg_len) { return (uint)mauve_default_seed_weight((int64_t)avg_len); }  // :511

typedef int score_t;
struct PairwiseScoringScheme {                   // progressiveMauve.cpp:666-687, repeatoire.cpp:1994
    score_t matrix[4][4];
    score_t gap_open, gap_extend;
    PairwiseScoringScheme()
    {
        mauve_scoring s; mauve_default_scoring(&s);
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) matrix[i][j] = s.matrix[i][j];
        gap_open = s.gap_open; gap_extend = s.gap_extend;
    }
    PairwiseScoringScheme(const score_t m[4][4], score_t go, score_t ge) : gap_open(go), gap_extend(ge)
    {
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) matrix[i][j] = m[i][j];
    }
};

// ---- AbstractMatch / Match: one length, signed 1-based starts (Appendix A of SURVEY.md) -----------------
class AbstractMatch {
public:
    enum orientation { forward, reverse, undefined };     // repeatoire.cpp:194, xmfa2maf.cpp:73
    virtual ~AbstractMatch() {}
};

class Match : public AbstractMatch {
public:
    explicit Match(uint seq_count = 0) : len_(0), start_(seq_count, NO_MATCH) {}
    uint SeqCount() const { return (uint)start_.size(); }
    gnSeqI Length(uint = 0) const { return (gnSeqI)len_; }
    gnSeqI AlignmentLength() const { return (gnSeqI)len_; }
    void SetLength(gnSeqI len) { len_ = (int64)len; }                       // SeedMatchEnumerator.h:76
    void SetLength(gnSeqI len, uint) { len_ = (int64)len; }                 // length first (repeatoire.cpp:241)
    int64 Start(uint seqI) const { return start_[seqI]; }
    int64 operator[](uint seqI) const { return start_[seqI]; }              // SeedMatchEnumerator.h:132
    void SetStart(uint seqI, int64 s) { start_[seqI] = s; }                 // :83 (sets left end and orientation)
    gnSeqI LeftEnd(uint seqI) const { return (gnSeqI)std::llabs(start_[seqI]); }       // scoreAlignment.cpp:165
    gnSeqI RightEnd(uint seqI) const { return LeftEnd(seqI) + (gnSeqI)len_ - 1; }      // :166
    void SetLeftEnd(uint seqI, gnSeqI pos) { start_[seqI] = start_[seqI] < 0 ? -(int64)pos : (int64)pos; }
    orientation Orientation(uint seqI) const { return start_[seqI] == NO_MATCH ? undefined : (start_[seqI] < 0 ? reverse : forward); }
    void SetOrientation(uint seqI, orientation o) { int64 a = std::llabs(start_[seqI]); start_[seqI] = o == reverse ? -a : a; }
    uint Multiplicity() const { uint m = 0; for (int64 s : start_) m += s != NO_MATCH; return m; }
    uint FirstStart() const { uint i = 0; while (i < start_.size() && start_[i] == NO_MATCH) i++; return i; }
    // column-unit crops (MatchRecord.h:263-276 uses the per-sequence forms on ungapped matches: same thing)
    void CropStart(gnSeqI cols) { crop(cols, 0); }
    void CropEnd(gnSeqI cols) { crop(0, cols); }
    void CropLeft(gnSeqI amount, uint seqI) { if (start_[seqI] > 0) crop(amount, 0); else crop(0, amount); }
    void CropRight(gnSeqI amount, uint seqI) { if (start_[seqI] > 0) crop(0, amount); else crop(amount, 0); }
    void Invert() { for (int64 &s : start_) s = -s; }                       // MatchRecord.h:283-284
    Match *Copy() const { return new Match(*this); }                        // SeedMatchEnumerator.h:119
    void Free() { delete this; }
private:
    void crop(gnSeqI first, gnSeqI last)
    {
        for (int64 &s : start_) { if (s > 0) s += (int64)first; else if (s < 0) s -= (int64)last; }
        len_ -= (int64)(first + last);
    }
    int64 len_;
    std::vector<int64> start_;
};
inline std::ostream &operator<<(std::ostream &os, const Match &m)          // row shape of MatchRecord.h:350-355
{
    os << m.Length();
    for (uint i = 0; i < m.SeqCount(); i++) os << '\t' << m.Start(i);
    return os;
}

// ---- device context shared by the objects of one alignment ----------------------------------------------
class HipContext {
public:
    explicit HipContext(int device = 0) : ctx_(nullptr)
    {
        if (mauve_ctx_create(device, &ctx_) != MAUVE_OK) throw genome::gnException(std::string("mauve_ctx_create: ") + mauve_last_error(nullptr));
    }
    ~HipContext() { mauve_ctx_destroy(ctx_); }
    mauve_ctx *get() const { return ctx_; }
    void check(int rc, const char *what) const
    {
        if (rc != MAUVE_OK) throw genome::gnException(std::string(what) + ": " + mauve_last_error(ctx_));
    }
    static HipContext &global() { static HipContext c(device_from_env()); return c; }
private:
    static int device_from_env() { const char *e = getenv("MAUVE_HIP_DEVICE"); return e ? atoi(e) : 0; }
    HipContext(const HipContext &); HipContext &operator=(const HipContext &);
    mauve_ctx *ctx_;
};

// ---- SortedMerList: the SML of one genome (GetMer(pos) 0-based, strand flag in bit 0) -------------------
class SortedMerList {
public:
    SortedMerList() : seed_(0), seq_index_(-1) {}
    virtual ~SortedMerList() {}
    uint SeedLength() const { return getSeedLength(seed_); }               // SeedMatchEnumerator.h:76
    uint SeedWeight() const { return (uint)mauve_seed_weight((uint64_t)seed_); }
    int64 Seed() const { return seed_; }
    gnSeqI Length() const { return mer_.size(); }
    // mer of the window starting at base `pos` (0-based): left-aligned mer | strand flag (SeedMatchEnumerator.h:133)
    uint64 GetMer(gnSeqI pos) const { return by_pos_.at((size_t)pos); }
    // i-th entry in sorted order
    uint64 SortedMer(gnSeqI i) const { return mer_.at((size_t)i); }
    gnSeqI SortedPosition(gnSeqI i) const { return (gnSeqI)pos_.at((size_t)i); }
    gnSeqI UniqueMerCount() const                                          // uniqueMerCount.cpp:39
    {
        gnSeqI n = 0;
        for (size_t i = 0; i < mer_.size(); i++) if (i == 0 || (mer_[i] >> 1) != (mer_[i - 1] >> 1)) n++;
        return n;
    }
    void Clear() { mer_.clear(); pos_.clear(); by_pos_.clear(); }
    // .sslist cache of a device-built list (DNAFileSML::LoadFile, uniqueMerCount.cpp:30-39; file naming
    // getDefaultSmlFileNames, progressiveMauve.cpp:215-224).  libMems' binary layout is [EXT]; this one is a small
    // little-endian header ("MHSSLIST", version, seed pattern, entries) followed by the mers and the positions.
    void WriteFile(const std::string &path) const
    {
        std::ofstream out(path.c_str(), std::ios::binary);
        if (!out) throw genome::gnException("SortedMerList::WriteFile: cannot open " + path);
        const char magic[8] = {'M', 'H', 'S', 'S', 'L', 'I', 'S', 'T'};
        const uint64_t hdr[3] = {1, (uint64_t)seed_, (uint64_t)mer_.size()};
        out.write(magic, 8); out.write((const char *)hdr, sizeof hdr);
        out.write((const char *)mer_.data(), (std::streamsize)(mer_.size() * 8));
        out.write((const char *)pos_.data(), (std::streamsize)(pos_.size() * 8));
        if (!out) throw genome::gnException("SortedMerList::WriteFile: write failed: " + path);
    }
    void LoadFile(const std::string &path)
    {
        std::ifstream in(path.c_str(), std::ios::binary);
        if (!in) throw genome::gnException("SortedMerList::LoadFile: cannot open " + path);
        char magic[8]; uint64_t hdr[3];
        in.read(magic, 8); in.read((char *)hdr, sizeof hdr);
        if (!in || std::string(magic, 8) != "MHSSLIST" || hdr[0] != 1) throw genome::gnException("SortedMerList::LoadFile: not a sorted mer list: " + path);
        seed_ = (int64)hdr[1]; seq_index_ = -1;
        mer_.assign((size_t)hdr[2], 0); pos_.assign((size_t)hdr[2], 0);
        in.read((char *)mer_.data(), (std::streamsize)(mer_.size() * 8));
        in.read((char *)pos_.data(), (std::streamsize)(pos_.size() * 8));
        if (!in) throw genome::gnException("SortedMerList::LoadFile: truncated file: " + path);
        by_pos_.assign(mer_.size(), 0);
        for (size_t i = 0; i < mer_.size(); i++) {
            if (pos_[i] < 0 || (size_t)pos_[i] >= mer_.size()) throw genome::gnException("SortedMerList::LoadFile: position out of range");
            by_pos_[(size_t)pos_[i]] = mer_[i];
        }
    }
    // filled by MatchList::CreateMemorySMLs
    void fill(HipContext &hc, int seq_index, int64 seed, gnSeqI seq_len)
    {
        seed_ = seed; seq_index_ = seq_index;
        int64_t span = mauve_seed_length((uint64_t)seed), n = (int64_t)seq_len - span + 1; if (n < 0) n = 0;
        mer_.assign((size_t)n, 0); pos_.assign((size_t)n, 0);
        int64_t got = 0;
        hc.check(mauve_sorted_mer_list(hc.get(), seq_index, (uint64_t)seed, mer_.data(), pos_.data(), &got), "mauve_sorted_mer_list");
        by_pos_.assign((size_t)n, 0);
        for (size_t i = 0; i < (size_t)got; i++) by_pos_[(size_t)pos_[i]] = mer_[i];
    }
private:
    int64 seed_; int seq_index_;
    std::vector<uint64_t> mer_; std::vector<int64_t> pos_; std::vector<uint64_t> by_pos_;
};

typedef SortedMerList DNAFileSML;                 // uniqueMerCount.cpp:30: the file-backed list is the same object here

// progressiveMauve.cpp:199-224: the seed pattern as a 0/1 string from its first set bit, and the default
// <sequence file>.<pattern>.sslist names
inline std::string getPatternText(int64 seed_pattern)
{
    std::string pat;
    for (int i = 63; i >= 0; i--) if (!pat.empty() || ((uint64)seed_pattern >> i & 1)) pat.push_back(((uint64)seed_pattern >> i & 1) ? '1' : '0');
    return pat;
}
inline void getDefaultSmlFileNames(const std::vector<std::string> &seq_files, std::vector<std::string> &sml_files, int seed_weight, int seed_rank)
{
    const std::string pattern = getPatternText(getSeed(seed_weight, seed_rank));
    sml_files.resize(seq_files.size());
    for (size_t i = 0; i < seq_files.size(); i++) sml_files[i] = seq_files[i] + "." + pattern + ".sslist";
}

// ---- MatchList: vector<Match*> + sequence and SML tables (mauveAligner.cpp:450-466,600,641-651) ----------
class MatchList : public std::vector<Match *> {
public:
    std::vector<genome::gnSequence *> seq_table;
    std::vector<SortedMerList *> sml_table;
    std::vector<std::string> seq_filename, sml_filename;
    int64 seed_pattern;
    MatchList() : seed_pattern(0) {}
    static uint GetDefaultMerSize(const std::vector<genome::gnSequence *> &seqs)     // mauveAligner.cpp:651
    {
        gnSeqI tot = 0; for (auto *s : seqs) tot += s->length();
        return seqs.empty() ? 0 : getDefaultSeedWeight(tot / seqs.size());
    }
    // uploads the sequences to the device and builds one SML per genome (mauveAligner.cpp:456)
    void CreateMemorySMLs(uint seed_weight, std::ostream *log = nullptr, int seed_rank = 0)
    {
        HipContext &hc = HipContext::global();
        upload(hc);
        if (seed_weight == 0) seed_weight = GetDefaultMerSize(seq_table);
        seed_pattern = getSeed((int)seed_weight, seed_rank);
        if (!seed_pattern) throw genome::gnException("CreateMemorySMLs: no seed for this weight/rank");
        for (auto *s : sml_table) delete s;
        sml_table.clear();
        for (size_t i = 0; i < seq_table.size(); i++) {
            SortedMerList *sml = new SortedMerList();
            sml->fill(hc, (int)i, seed_pattern, seq_table[i]->length());
            sml_table.push_back(sml);
            if (log) *log << "Sorted mer list " << i << ": " << sml->Length() << " mers\n";
        }
    }
    void LoadSMLs(uint seed_weight, std::ostream *log = nullptr, int seed_rank = 0) { CreateMemorySMLs(seed_weight, log, seed_rank); }
    void upload(HipContext &hc) const
    {
        std::vector<std::vector<uint64_t>> packed(seq_table.size());
        std::vector<const uint64_t *> ptr; std::vector<int64_t> lens;
        for (size_t i = 0; i < seq_table.size(); i++) {
            const std::string &s = seq_table[i]->str();
            packed[i].assign(mauve_packed_words((int64_t)s.size()), 0);
            mauve_pack_ascii(s.data(), (int64_t)s.size(), packed[i].data());
            ptr.push_back(packed[i].data()); lens.push_back((int64_t)s.size());
        }
        hc.check(mauve_set_genomes(hc.get(), (int)seq_table.size(), ptr.data(), lens.data()), "mauve_set_genomes");
    }
    void MultiplicityFilter(uint mult)                                     // mauveAligner.cpp:600
    {
        size_t k = 0;
        for (size_t i = 0; i < size(); i++) { if ((*this)[i]->Multiplicity() == mult) (*this)[k++] = (*this)[i]; else (*this)[i]->Free(); }
        resize(k);
    }
    void Clear() { for (Match *m : *this) m->Free(); clear(); }            // MLDeleter, mauveAligner.cpp:39-45
};

// ---- MatchFinder family ----------------------------------------------------------------------------------
class MatchFinder {
public:
    MatchFinder() : seq_count(0), mask_(0), log_(nullptr) {}
    virtual ~MatchFinder() {}
    virtual MatchFinder *Clone() const = 0;                                // UniqueMatchFinder.h:28
    boolean AddSequence(SortedMerList *sar, genome::gnSequence *seq)       // SeedMatchEnumerator.h:25
    {
        if (!sar || !seq) return false;
        if (!sar_table.empty() && sar->Seed() != sar_table[0]->Seed()) return false;
        sar_table.push_back(sar); seq_table_.push_back(seq); seq_count++;
        return true;
    }
    void LogProgress(std::ostream *os) { log_ = os; }                      // mauveAligner.cpp:532
    void SetMask(uint64 m) { mask_ = m; }                                  // mauveAligner.cpp:530 (MaskedMemHash)
    void ClearSequences() { sar_table.clear(); seq_table_.clear(); seq_count = 0; }
    virtual void Clear() { found_len_.clear(); found_start_.clear(); }
    // progressiveMauve.cpp:490-501: finder.FindMatches(match_list)
    virtual void FindMatches(MatchList &ml)
    {
        ClearSequences();
        for (size_t i = 0; i < ml.seq_table.size(); i++)
            if (!AddSequence(ml.sml_table.at(i), ml.seq_table[i])) { genome::ErrorMsg("Error adding " + (i < ml.seq_filename.size() ? ml.seq_filename[i] : std::string("sequence")) + "\n"); return; }
        CreateMatches();
        GetMatchList(ml);
    }
    virtual boolean CreateMatches()
    {
        if (seq_count < 1) return false;
        HipContext &hc = HipContext::global();
        int64_t n = 0;
        hc.check(mauve_seed_mums(hc.get(), (uint64_t)sar_table[0]->Seed(), seedRule(), mask_, extendMatches() ? 1 : 0, &n), "mauve_seed_mums");
        found_len_.assign((size_t)n, 0); found_start_.assign((size_t)n * seq_count, 0);
        hc.check(mauve_get_matches(hc.get(), found_len_.data(), found_start_.data()), "mauve_get_matches");
        if (log_) *log_ << "100%..done, " << n << " matches\n";
        return true;
    }
    void GetMatchList(MatchList &ml) const                                 // progressiveMauve.cpp:545
    {
        for (size_t i = 0; i < found_len_.size(); i++) {
            Match *m = new Match(seq_count);
            m->SetLength((gnSeqI)found_len_[i]);
            for (uint g = 0; g < seq_count; g++) m->SetStart(g, found_start_[i * seq_count + g]);
            ml.push_back(m);
        }
    }
    virtual SortedMerList *GetSar(uint32 sarI) const { return sar_table.at(sarI); }    // SeedMatchEnumerator.h:40
protected:
    virtual int seedRule() const = 0;            // which EnumerateMatches rule runs in the join kernel
    virtual bool extendMatches() const { return true; }
    uint seq_count;                              // SeedMatchEnumerator.h:60
    std::vector<SortedMerList *> sar_table;      // :56
    std::vector<genome::gnSequence *> seq_table_;
    uint64 mask_;
    std::ostream *log_;
    std::vector<int64_t> found_len_, found_start_;
};

class MemHash : public MatchFinder {             // mauveAligner.cpp:523 (via MaskedMemHash), UniqueMatchFinder.h:21
public:
    virtual MemHash *Clone() const { return new MemHash(*this); }
protected:
    virtual int seedRule() const { return MAUVE_MODE_MEM; }
};
class MaskedMemHash : public MemHash {
public:
    virtual MaskedMemHash *Clone() const { return new MaskedMemHash(*this); }
};
// progressiveMauve.cpp:496-501: "PairwiseMatchFinder pmf; pmf.FindMatches(pairwise_match_list)" for <= 4 genomes
class PairwiseMatchFinder : public MemHash {
public:
    virtual PairwiseMatchFinder *Clone() const { return new PairwiseMatchFinder(*this); }
protected:
    virtual int seedRule() const { return MAUVE_MODE_PAIRWISE; }
};

}  // namespace mems

// In-tree subclasses, same names and scope as the reference (global namespace).
class UniqueMatchFinder : public mems::MemHash {            // src/UniqueMatchFinder.h:21-32
public:
    UniqueMatchFinder() {}
    ~UniqueMatchFinder() {}
    UniqueMatchFinder(const UniqueMatchFinder &mh) : mems::MemHash(mh) {}
    virtual UniqueMatchFinder *Clone() const { return new UniqueMatchFinder(*this); }
protected:
    virtual int seedRule() const { return MAUVE_MODE_UNIQUE; }   // UniqueMatchFinder.cpp:36-60 in the join kernel
};

class SeedMatchEnumerator : public mems::MatchFinder {      // src/SeedMatchEnumerator.h:14-49
public:
    virtual SeedMatchEnumerator *Clone() const { return new SeedMatchEnumerator(*this); }
    // SeedMatchEnumerator.h:19-33: single genome, every repeated seed becomes a Match of seed length
    void FindMatches(mems::MatchList &match_list, size_t min_multi = 2, size_t max_multi = 1000, bool direct_repeats_only = false)
    {
        ClearSequences();
        for (size_t seqI = 0; seqI < match_list.seq_table.size(); ++seqI)
            if (!AddSequence(match_list.sml_table[seqI], match_list.seq_table[seqI])) {
                genome::ErrorMsg("Error adding " + (seqI < match_list.seq_filename.size() ? match_list.seq_filename[seqI] : std::string("sequence")) + "\n");
                return;
            }
        match_list.clear();
        if (seq_count != 1) return;                                         // CreateMatches, :59-65
        mems::HipContext &hc = mems::HipContext::global();
        int64_t n = 0, ns = 0;
        const uint64_t pat = (uint64_t)sar_table[0]->Seed();
        hc.check(mauve_seed_match_enumerate(hc.get(), 0, pat, (int64_t)min_multi, (int64_t)max_multi, direct_repeats_only, &n, &ns, nullptr, nullptr, nullptr), "mauve_seed_match_enumerate");
        std::vector<int64_t> mult((size_t)n), off((size_t)n + 1), st((size_t)ns);
        hc.check(mauve_seed_match_enumerate(hc.get(), 0, pat, (int64_t)min_multi, (int64_t)max_multi, direct_repeats_only, &n, &ns, mult.data(), off.data(), st.data()), "mauve_seed_match_enumerate");
        for (int64_t i = 0; i < n; i++) {
            mems::Match *m = new mems::Match((uint)mult[(size_t)i]);
            m->SetLength(GetSar(0)->SeedLength());
            for (int64_t k = 0; k < mult[(size_t)i]; k++) m->SetStart((uint)k, st[(size_t)(off[(size_t)i] + k)]);
            match_list.push_back(m);
        }
    }
    virtual mems::SortedMerList *GetSar(uint32) const { return sar_table[0]; }   // :54-57
protected:
    virtual int seedRule() const { return MAUVE_MODE_MEM; }
    virtual bool extendMatches() const { return false; }
};

namespace mems {

// ---- gapped alignment seam --------------------------------------------------------------------------------
class GappedAligner {                            // Aligner::SetGappedAligner(GappedAligner&), mauveAligner.cpp:674
public:
    virtual ~GappedAligner() {}
    // CallMuscleFast shape (repeatoire.cpp:1262): aligned rows out, raw sequences in
    virtual bool CallMuscleFast(std::vector<std::string> &aln_out, const std::vector<std::string> &seqs_in, int gap_open, int gap_extend) = 0;
};

class HipGappedAligner : public GappedAligner {  // stands where MuscleInterface::getMuscleInterface() stood (:82)
public:
    static HipGappedAligner &getInterface() { static HipGappedAligner g; return g; }
    void SetScoring(const PairwiseScoringScheme &p) { pss_ = p; }
    virtual bool CallMuscleFast(std::vector<std::string> &aln_out, const std::vector<std::string> &seqs_in, int gap_open, int gap_extend)
    {
        const int N = (int)seqs_in.size();
        if (N < 1 || N > MAUVE_MAX_SEQ) return false;
        std::vector<uint8_t> codes; std::vector<int64_t> off(1, 0);
        for (const std::string &s : seqs_in) {
            for (char ch : s) codes.push_back(ch == 'C' || ch == 'c' ? 1 : ch == 'G' || ch == 'g' ? 2 : ch == 'T' || ch == 't' ? 3 : 0);
            off.push_back((int64_t)codes.size());
        }
        mauve_scoring sc;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) sc.matrix[i][j] = pss_.matrix[i][j];
        sc.gap_open = gap_open; sc.gap_extend = gap_extend;
        std::vector<uint32_t> cols(codes.size() + 1); int64_t col_off[2] = {0, 0}, score = 0;
        HipContext &hc = HipContext::global();
        if (mauve_dp_batch(hc.get(), N, 1, codes.empty() ? nullptr : codes.data(), off.data(), &sc, cols.data(), col_off, &score) != MAUVE_OK) return false;
        aln_out.assign((size_t)N, std::string());
        std::vector<size_t> nxt((size_t)N, 0);
        for (int64_t c = 0; c < col_off[1]; c++)
            for (int g = 0; g < N; g++) aln_out[(size_t)g].push_back((cols[(size_t)c] >> g & 1) ? seqs_in[(size_t)g][nxt[(size_t)g]++] : '-');
        return true;
    }
private:
    PairwiseScoringScheme pss_;
};

// ---- Interval / IntervalList: the result of Aligner::align, resident on the host ------------------------------
// One Interval is one block of the alignment: for every genome its range and strand (absent: left = right = 0)
// and, per alignment column, the set of genomes that have a base there.  The bases themselves stay in seq_table.
class Interval {
public:
    Interval() {}
    Interval(const std::vector<int64> &left, const std::vector<int64> &right, const std::vector<char> &reverse,
             const std::vector<uint32_t> &cols) : left_(left), right_(right), rev_(reverse), cols_(cols) {}
    uint SeqCount() const { return (uint)left_.size(); }
    gnSeqI LeftEnd(uint seqI) const { return (gnSeqI)left_[seqI]; }                       // toGrimmFormat.cpp:62-77
    gnSeqI RightEnd(uint seqI) const { return (gnSeqI)right_[seqI]; }
    gnSeqI Length(uint seqI) const { return left_[seqI] ? (gnSeqI)(right_[seqI] - left_[seqI] + 1) : 0; }
    int64 Start(uint seqI) const { return rev_[seqI] ? -left_[seqI] : left_[seqI]; }      // signed, NO_MATCH when absent
    AbstractMatch::orientation Orientation(uint seqI) const
    { return left_[seqI] == NO_MATCH ? AbstractMatch::undefined : (rev_[seqI] ? AbstractMatch::reverse : AbstractMatch::forward); }
    uint Multiplicity() const { uint m = 0; for (int64 l : left_) m += l != NO_MATCH; return m; }
    gnSeqI AlignmentLength() const { return (gnSeqI)cols_.size(); }
    const std::vector<uint32_t> &Columns() const { return cols_; }
    // rows of the block as '-'-gapped strings, one per genome (all gaps for an absent genome); a reverse
    // component is written as the reverse complement (GetAlignment, repeatoire.cpp:1264-1265)
    void GetAlignment(std::vector<std::string> &rows, const std::vector<genome::gnSequence *> &seq_table) const
    {
        const uint N = SeqCount();
        rows.assign(N, std::string(cols_.size(), '-'));
        for (uint g = 0; g < N; g++) {
            if (!left_[g]) continue;
            if (g >= seq_table.size() || (gnSeqI)right_[g] > seq_table[g]->length()) throw genome::gnException("Interval::GetAlignment: sequence table does not cover the interval");
            const std::string &sq = seq_table[g]->str();
            int64 nxt = rev_[g] ? right_[g] : left_[g];
            for (size_t k = 0; k < cols_.size(); k++) {
                if (!(cols_[k] >> g & 1)) continue;
                rows[g][k] = base_char(sq[(size_t)nxt - 1], rev_[g] != 0);
                nxt += rev_[g] ? -1 : 1;
            }
        }
    }
    // the letter the device path sees: upper case ACGT, everything else reads as A (mauve_pack_ascii)
    static char base_char(char c, bool complement)
    {
        int code;
        switch (c) { case 'C': case 'c': code = 1; break; case 'G': case 'g': code = 2; break; case 'T': case 't': code = 3; break; default: code = 0; }
        return "ACGT"[complement ? 3 - code : code];
    }
private:
    std::vector<int64> left_, right_;
    std::vector<char> rev_;
    std::vector<uint32_t> cols_;
};

class IntervalList : public std::vector<Interval> {
public:
    std::vector<genome::gnSequence *> seq_table;
    std::vector<std::string> seq_filename;
    mauve_align_sizes sizes;
    IntervalList() { sizes = mauve_align_sizes(); }

    // pull the interval table of the last mauve_align / mauve_progressive_align off the context
    void fetch(HipContext &hc, uint seq_count)
    {
        clear();
        const size_t K = (size_t)sizes.n_iv, N = seq_count;
        std::vector<int64_t> left(K * N), right(K * N), col_off(K + 1);
        std::vector<int8_t> rev(K * N);
        std::vector<uint32_t> cols((size_t)sizes.n_cols);
        hc.check(mauve_align_fetch(hc.get(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, left.data(),
                                   right.data(), rev.data(), col_off.data(), cols.data(), nullptr), "mauve_align_fetch");
        for (size_t i = 0; i < K; i++) {
            std::vector<int64> l(left.begin() + i * N, left.begin() + (i + 1) * N), r(right.begin() + i * N, right.begin() + (i + 1) * N);
            std::vector<char> rv(rev.begin() + i * N, rev.begin() + (i + 1) * N);
            push_back(Interval(l, r, rv, std::vector<uint32_t>(cols.begin() + col_off[i], cols.begin() + col_off[i + 1])));
        }
    }

    // XMFA (mauveAligner.cpp:746-760; layout mfa2xmfa.cpp:64-115): byte-identical to mauve_write_xmfa
    void WriteStandardAlignment(std::ostream &os) const
    {
        const uint N = (uint)seq_table.size();
        os << "#FormatVersion Mauve1\n";
        for (uint g = 0; g < N; g++)
            os << "#Sequence" << g + 1 << "File\t" << name(g) << "\n#Sequence" << g + 1 << "Entry\t" << g + 1 << "\n#Sequence" << g + 1
               << "Format\tFastA\n";
        std::vector<std::string> rows;
        for (const Interval &iv : *this) {
            iv.GetAlignment(rows, seq_table);
            for (uint g = 0; g < N && g < iv.SeqCount(); g++) {
                if (!iv.LeftEnd(g)) continue;
                os << "> " << g + 1 << ':' << iv.LeftEnd(g) << '-' << iv.RightEnd(g) << ' ' << (iv.Orientation(g) == AbstractMatch::reverse ? '-' : '+')
                   << ' ' << name(g) << '\n';
                for (size_t pos = 0; pos < rows[g].size(); pos += 80) os << rows[g].substr(pos, 80) << '\n';
            }
            os << "=\n";
        }
    }

    // XMFA reader (ReadStandardAlignment, scoreProcrastAlignment.cpp:442): ranges, strands and gap pattern of every
    // block; the number of genomes comes from the header (or the largest index seen).  seq_table is left alone:
    // as with libMems the caller loads the sequences named in seq_filename.
    void ReadStandardAlignment(std::istream &is)
    {
        clear(); seq_filename.clear();
        std::string line;
        struct Row { uint g; int64 lo, hi; bool rev; std::string txt; };
        std::vector<std::vector<Row>> blocks(1);
        uint N = 0;
        while (std::getline(is, line)) {
            if (!line.empty() && line[line.size() - 1] == '\r') line.erase(line.size() - 1);
            if (line.empty()) continue;
            if (line[0] == '#') {
                unsigned idx = 0; char tag[32];
                if (sscanf(line.c_str(), "#Sequence%u%31[A-Za-z]", &idx, tag) == 2 && std::string(tag) == "File" && idx >= 1) {
                    if (seq_filename.size() < idx) seq_filename.resize(idx);
                    const size_t tab = line.find('\t');
                    seq_filename[idx - 1] = tab == std::string::npos ? "" : line.substr(tab + 1);
                    N = std::max(N, (uint)idx);
                }
                continue;
            }
            if (line[0] == '=') { blocks.push_back(std::vector<Row>()); continue; }
            if (line[0] == '>') {
                Row r; unsigned g = 0; long long lo = 0, hi = 0; char strand = '+';
                if (sscanf(line.c_str(), "> %u:%lld-%lld %c", &g, &lo, &hi, &strand) < 3 || g < 1) throw genome::gnException("ReadStandardAlignment: bad defline: " + line);
                r.g = g - 1; r.lo = lo; r.hi = hi; r.rev = strand == '-';
                N = std::max(N, (uint)g);
                blocks.back().push_back(r);
                continue;
            }
            if (blocks.back().empty()) throw genome::gnException("ReadStandardAlignment: sequence data before a defline");
            blocks.back().back().txt += line;
        }
        for (const auto &blk : blocks) {
            if (blk.empty()) continue;
            const size_t len = blk[0].txt.size();
            std::vector<int64> l(N, 0), r(N, 0); std::vector<char> rv(N, 0); std::vector<uint32_t> cols(len, 0);
            for (const Row &row : blk) {
                if (row.txt.size() != len) throw genome::gnException("ReadStandardAlignment: ragged block");
                if (row.lo == 0 && row.hi == 0) continue;            // some writers list absent genomes as 0-0
                l[row.g] = row.lo; r[row.g] = row.hi; rv[row.g] = row.rev;
                int64 bases = 0;
                for (size_t k = 0; k < len; k++) if (row.txt[k] != '-') { cols[k] |= 1u << row.g; bases++; }
                if (bases != row.hi - row.lo + 1) throw genome::gnException("ReadStandardAlignment: range and residue count disagree");
            }
            push_back(Interval(l, r, rv, cols));
        }
        sizes = mauve_align_sizes(); sizes.n_iv = (int64_t)size();
        for (const Interval &iv : *this) sizes.n_cols += (int64_t)iv.AlignmentLength();
    }

    // .mln (mauveAligner.cpp:702,715).  libMems owns the real layout [EXT]; this one is self-describing text in the
    // same spirit (tab-separated header, then per interval the signed starts, lengths and the gap pattern
    // run-length encoded per genome), and round-trips through ReadList.
    void WriteList(std::ostream &os) const
    {
        const uint N = (uint)seq_table.size();
        os << "FormatVersion\t4\nSequenceCount\t" << N << '\n';
        for (uint g = 0; g < N; g++) os << "Sequence" << g << "File\t" << name(g) << "\nSequence" << g << "Length\t" << seq_table[g]->length() << '\n';
        os << "IntervalCount\t" << size() << '\n';
        for (size_t i = 0; i < size(); i++) {
            const Interval &iv = (*this)[i];
            os << "Interval\t" << i << '\t' << iv.AlignmentLength() << '\n';
            for (uint g = 0; g < iv.SeqCount(); g++) {
                os << iv.Start(g) << '\t' << iv.Length(g);
                // runs: +n = n columns with a base, -n = n gap columns
                const std::vector<uint32_t> &c = iv.Columns();
                for (size_t k = 0; k < c.size();) {
                    const bool on = c[k] >> g & 1; size_t j = k;
                    while (j < c.size() && ((c[j] >> g & 1) != 0) == on) j++;
                    os << '\t' << (on ? "" : "-") << (j - k);
                    k = j;
                }
                os << '\n';
            }
        }
    }
    void ReadList(std::istream &is)
    {
        clear(); seq_filename.clear();
        std::string key; uint N = 0; size_t K = 0; std::string line;
        auto expect = [&](const std::string &k) { if (!(is >> key) || key != k) throw genome::gnException("IntervalList::ReadList: expected " + k); };
        expect("FormatVersion"); int ver; is >> ver;
        expect("SequenceCount"); is >> N;
        for (uint g = 0; g < N; g++) {
            is >> key; std::getline(is, line); seq_filename.push_back(line.empty() ? "" : line.substr(1));
            is >> key; long long len; is >> len;
        }
        expect("IntervalCount"); is >> K;
        for (size_t i = 0; i < K; i++) {
            size_t idx, alen; expect("Interval"); is >> idx >> alen;
            std::getline(is, line);
            std::vector<int64> l(N, 0), r(N, 0); std::vector<char> rv(N, 0); std::vector<uint32_t> cols(alen, 0);
            for (uint g = 0; g < N; g++) {
                if (!std::getline(is, line)) throw genome::gnException("IntervalList::ReadList: truncated interval");
                std::istringstream ls(line);
                long long st, len; ls >> st >> len;
                l[g] = std::llabs(st); r[g] = st ? l[g] + len - 1 : 0; rv[g] = st < 0;
                long long run; size_t k = 0;
                while (ls >> run) {
                    const size_t n = (size_t)std::llabs(run);
                    if (k + n > alen) throw genome::gnException("IntervalList::ReadList: runs exceed the alignment length");
                    if (run > 0) for (size_t j = 0; j < n; j++) cols[k + j] |= 1u << g;
                    k += n;
                }
            }
            push_back(Interval(l, r, rv, cols));
        }
        sizes = mauve_align_sizes(); sizes.n_iv = (int64_t)size();
        for (const Interval &iv : *this) sizes.n_cols += (int64_t)iv.AlignmentLength();
    }
private:
    std::string name(uint g) const { return g < seq_filename.size() ? seq_filename[g] : std::string(); }
};

// ---- .mums: the match list at the seam between the seed stage and the aligner (mauveAligner.cpp:484,499,603;
// progressiveMauve.cpp:476,552).  Header as libMems writes it [EXT, from Mauve's published files]: FormatVersion,
// SequenceCount, Sequence<i>File / Sequence<i>Length, MatchCount; then one row per match in the layout of
// operator<< above (length, signed starts; MatchRecord.h:350-355 prints the same row).
inline void WriteList(const MatchList &ml, std::ostream &os)
{
    os << "FormatVersion\t3\nSequenceCount\t" << ml.seq_table.size() << '\n';
    for (size_t g = 0; g < ml.seq_table.size(); g++)
        os << "Sequence" << g << "File\t" << (g < ml.seq_filename.size() ? ml.seq_filename[g] : std::string()) << "\nSequence" << g << "Length\t"
           << ml.seq_table[g]->length() << '\n';
    os << "MatchCount\t" << ml.size() << '\n';
    for (const Match *m : ml) os << *m << '\n';
}
inline void ReadList(MatchList &ml, std::istream &is)
{
    ml.Clear(); ml.seq_filename.clear();
    std::string key, line; size_t N = 0, M = 0; int ver = 0;
    auto expect = [&](const std::string &k) { if (!(is >> key) || key != k) throw genome::gnException("ReadList: expected " + k); };
    expect("FormatVersion"); is >> ver;
    expect("SequenceCount"); is >> N;
    for (size_t g = 0; g < N; g++) {
        is >> key; std::getline(is, line); ml.seq_filename.push_back(line.empty() ? "" : line.substr(1));
        long long len; is >> key >> len;
    }
    expect("MatchCount"); is >> M;
    for (size_t i = 0; i < M; i++) {
        long long len, st;
        if (!(is >> len)) throw genome::gnException("ReadList: truncated match list");
        Match *m = new Match((uint)N);
        m->SetLength((gnSeqI)len);
        for (size_t g = 0; g < N; g++) { if (!(is >> st)) { m->Free(); throw genome::gnException("ReadList: truncated match row"); } m->SetStart((uint)g, st); }
        ml.push_back(m);
    }
}

// ---- Aligner: setters and align() as called at mauveAligner.cpp:668-698 -----------------------------------
class Aligner {
public:
    explicit Aligner(uint seq_count) : seq_count_(seq_count) { mauve_default_params(&p_); }
    void SetMinRecursionGapLength(gnSeqI n) { p_.min_recursive_gap = (int64_t)n; }     // :670-672
    void SetGappedAligner(GappedAligner &) {}                                           // :674 (the HIP DP is built in)
    void SetMaxGappedAlignmentLength(gnSeqI n) { p_.max_gapped_len = (int64_t)n; }     // :675-676
    void SetMaxExtensionIterations(uint n) { p_.max_extension_iters = (int32_t)n; }      // :687-690 (LCB extension)
    void SetSeedPattern(int64 seed) { p_.seed_pattern = (uint64_t)seed; }
    void SetScoring(const PairwiseScoringScheme &pss)
    {
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) p_.scoring.matrix[i][j] = pss.matrix[i][j];
        p_.scoring.gap_open = pss.gap_open; p_.scoring.gap_extend = pss.gap_extend;
    }
    // align(match_list, interval_list, 0, LCB_size, recursive, extend_lcbs, gapped, tree_filename)  (:698)
    // LCB_size < 0 = collinear hack (:665-666).  The multi-MUM search is redone on the device from the seed
    // pattern of match_list (it is the same search FindMatches performed), so the anchors never leave HBM.
    void align(MatchList &ml, IntervalList &il, double, int64 LCB_size, boolean recursive, boolean extend_lcbs, boolean gapped, std::string = "")
    {
        HipContext &hc = HipContext::global();
        if (ml.seq_table.size() != seq_count_) throw genome::gnException("Aligner::align: sequence count mismatch");
        ml.upload(hc);
        mauve_params p = p_;
        if (!p.seed_pattern) p.seed_pattern = (uint64_t)ml.seed_pattern;
        p.collinear = LCB_size < 0; p.lcb_weight = LCB_size < 0 ? -1 : LCB_size;
        p.recursive = recursive; p.gapped = gapped; p.extend_lcbs = extend_lcbs;
        hc.check(mauve_align(hc.get(), &p, &il.sizes), "mauve_align");
        il.seq_table = ml.seq_table; il.seq_filename = ml.seq_filename;
        il.fetch(hc, seq_count_);
    }
private:
    uint seq_count_;
    mauve_params p_;
};

// ---- ProgressiveAligner: setters and align() as called at progressiveMauve.cpp:575-710 -------------------
// The guide tree and the progressive anchoring run on the device (mauve_progressive_align).  The
// setters that tune libMems' sum-of-pairs LCB scoring have no counterpart in the frozen replacement; they are
// accepted so that the call site compiles, and documented as inert.
class ProgressiveAligner {
public:
    explicit ProgressiveAligner(uint seq_count) : seq_count_(seq_count), tree_left_(2 * seq_count - 1, -1), tree_right_(2 * seq_count - 1, -1)
    {
        mauve_default_params(&p_);
    }
    void setBreakpointPenalty(double w) { if (w >= 0) p_.lcb_weight = (int64_t)w * (int64_t)seq_count_; }   // --weight, :584-593
    void setMinimumBreakpointPenalty(double) {}
    void setCollinear(boolean c) { p_.collinear = c; }                        // :594-597
    void setGappedAlignment(boolean g) { p_.gapped = g; }                     // --skip-gapped-alignment
    void setRefinement(boolean) {}                                            // :578-579 (no refinement stage)
    void setRecursion(boolean r) { p_.recursive = r; }                        // :661-664
    void SetMaxGappedAlignmentLength(gnSeqI n) { p_.max_gapped_len = (int64_t)n; }
    void setPairwiseScoringScheme(const PairwiseScoringScheme &pss)           // :666-687
    {
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) p_.scoring.matrix[i][j] = pss.matrix[i][j];
        p_.scoring.gap_open = pss.gap_open; p_.scoring.gap_extend = pss.gap_extend;
    }
    void setLcbScoringScheme(int) {}                                          // :611-625 (ExtantSumOfPairs...: inert)
    void setUseLcbWeightScaling(boolean) {}                                   // :626-642
    void setBpDistEstimateMinScore(double) {}
    void setUseSeedFamilies(boolean) {}
    void SetUseCacheDb(boolean) {}                                            // :643-646
    // progressiveMauve.cpp:652-655 hands the pairwise matches over; the device path finds them itself
    // (PairwiseMatchFinder rule on the resident genomes), so only the seed pattern is taken from the list.
    void setPairwiseMatches(MatchList &pairwise) { if (pairwise.seed_pattern) p_.seed_pattern = (uint64_t)pairwise.seed_pattern; }
    void setSeedWeight(uint w) { p_.seed_weight = (int32_t)w; }
    // aligner.align(interval_list.seq_table, interval_list)  (:710)
    void align(std::vector<genome::gnSequence *> &seq_table, IntervalList &il)
    {
        if (seq_table.size() != seq_count_) throw genome::gnException("ProgressiveAligner::align: sequence count mismatch");
        HipContext &hc = HipContext::global();
        MatchList tmp; tmp.seq_table = seq_table;
        tmp.upload(hc);
        hc.check(mauve_progressive_align(hc.get(), &p_, &il.sizes, tree_left_.data(), tree_right_.data(), nullptr), "mauve_progressive_align");
        il.seq_table = seq_table;
        il.fetch(hc, seq_count_);
    }
    // guide tree of the last align(): child ids per node (leaves -1), nodes seq_count.. in merge order
    const std::vector<int32_t> &treeLeft() const { return tree_left_; }
    const std::vector<int32_t> &treeRight() const { return tree_right_; }
private:
    uint seq_count_;
    mauve_params p_;
    std::vector<int32_t> tree_left_, tree_right_;
};

}  // namespace mems
#endif
