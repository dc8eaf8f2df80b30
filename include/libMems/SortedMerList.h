// libMems/SortedMerList.h -- the sorted mer list (SML) of one genome as the in-tree code reads it: SeedLength()
// (SeedMatchEnumerator.h:76), GetMer(pos) with the strand flag of the canonical mer in bit 0 (:133),
// UniqueMerCount() (uniqueMerCount.cpp:39), plus the seed helpers getSeed / getSeedLength / getDefaultSeedWeight
// (progressiveMauve.cpp:217,511-517) and the device context the mirror classes share.
// The list itself is built by the device (mauve_sorted_mer_list) and only comes to the host when something reads
// it entry by entry: the in-kernel finders never need it.
#ifndef MAUVE_HIP_SORTEDMERLIST_H
#define MAUVE_HIP_SORTEDMERLIST_H

#include <algorithm>
#include "AbstractMatch.h"

namespace mems {

static const int CODING_SEED = MAUVE_CODING_SEED;
static const int SOLID_SEED = MAUVE_SOLID_SEED;  // repeatoire.cpp:1847

inline int64 getSeed(int weight, int rank = 0) { return (int64)mauve_get_seed(weight, rank); }       // progressiveMauve.cpp:217
inline uint getSeedLength(int64 seed) { return (uint)mauve_seed_length((uint64_t)seed); }            // :515-517
inline uint getDefaultSeedWeight(gnSeqI avg_len) { return (uint)mauve_default_seed_weight((int64_t)avg_len); }  // :511

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

// ---- SortedMerList ---------------------------------------------------------------------------------------
class SortedMerList {
public:
    SortedMerList() : seed_(0), seq_index_(-1), seq_len_(0), hc_(nullptr), loaded_(true) {}
    virtual ~SortedMerList() {}
    uint SeedLength() const { return getSeedLength(seed_); }               // SeedMatchEnumerator.h:76
    uint SeedWeight() const { return (uint)mauve_seed_weight((uint64_t)seed_); }
    int64 Seed() const { return seed_; }
    gnSeqI Length() const { if (!loaded_) { const int64_t n = (int64_t)seq_len_ - (int64_t)SeedLength() + 1; return n > 0 ? (gnSeqI)n : 0; } return mer_.size(); }
    // mer of the window starting at base `pos` (0-based): left-aligned mer | strand flag (SeedMatchEnumerator.h:133)
    uint64 GetMer(gnSeqI pos) const { load(); return by_pos_.at((size_t)pos); }
    // i-th entry in sorted order
    uint64 SortedMer(gnSeqI i) const { load(); return mer_.at((size_t)i); }
    gnSeqI SortedPosition(gnSeqI i) const { load(); return (gnSeqI)pos_.at((size_t)i); }
    gnSeqI UniqueMerCount() const                                          // uniqueMerCount.cpp:39
    {
        load();
        gnSeqI n = 0;
        for (size_t i = 0; i < mer_.size(); i++) if (i == 0 || (mer_[i] >> 1) != (mer_[i - 1] >> 1)) n++;
        return n;
    }
    void Clear() { mer_.clear(); pos_.clear(); by_pos_.clear(); loaded_ = true; hc_ = nullptr; }
    // .sslist cache of a device-built list (DNAFileSML::LoadFile, uniqueMerCount.cpp:30-39; file naming
    // getDefaultSmlFileNames, progressiveMauve.cpp:215-224).  libMems' binary layout is not reproduced: this is a
    // layout of its own, tagged "MHSSLIST" so that neither side mistakes the other's files (LoadFile rejects
    // anything without the tag): little-endian header (tag, version, seed pattern, entries), the mers, the positions.
    void WriteFile(const std::string &path) const
    {
        load();
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
        if (!in || std::string(magic, 8) != "MHSSLIST" || hdr[0] != 1)
            throw genome::gnException("SortedMerList::LoadFile: not a sorted mer list written by this library (libMems' own .sslist layout is not read): " + path);
        seed_ = (int64)hdr[1]; seq_index_ = -1; hc_ = nullptr; loaded_ = true;
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
    // set by MatchList::CreateMemorySMLs: which resident genome, which pattern; the entries follow on first use
    void fill(HipContext &hc, int seq_index, int64 seed, gnSeqI seq_len)
    {
        seed_ = seed; seq_index_ = seq_index; seq_len_ = seq_len; hc_ = &hc; loaded_ = false;
        mer_.clear(); pos_.clear(); by_pos_.clear();
    }
    int SequenceIndex() const { return seq_index_; }
private:
    void load() const
    {
        if (loaded_) return;
        int64_t span = mauve_seed_length((uint64_t)seed_), n = (int64_t)seq_len_ - span + 1; if (n < 0) n = 0;
        mer_.assign((size_t)n, 0); pos_.assign((size_t)n, 0);
        int64_t got = 0;
        hc_->check(mauve_sorted_mer_list(hc_->get(), seq_index_, (uint64_t)seed_, mer_.data(), pos_.data(), &got), "mauve_sorted_mer_list");
        by_pos_.assign((size_t)n, 0);
        for (size_t i = 0; i < (size_t)got; i++) by_pos_[(size_t)pos_[i]] = mer_[i];
        loaded_ = true;
    }
    int64 seed_; int seq_index_; gnSeqI seq_len_; HipContext *hc_;
    mutable bool loaded_;
    mutable std::vector<uint64_t> mer_; mutable std::vector<int64_t> pos_; mutable std::vector<uint64_t> by_pos_;
};

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

}  // namespace mems
#endif
