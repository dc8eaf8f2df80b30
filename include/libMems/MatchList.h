// libMems/MatchList.h -- vector<Match*> plus the sequence and SML tables (mauveAligner.cpp:450-466,600,641-651), and the
// .mums text format at the seam between the seed stage and the aligner (ReadList / WriteList, :484,499,603).
#ifndef MAUVE_HIP_MATCHLIST_H
#define MAUVE_HIP_MATCHLIST_H

#include <fstream>
#include <sstream>
#include "Match.h"
#include "MatchProjectionAdapter.h"      // in-tree code reaches it through this header (SeedMatchEnumerator.h:98)
#include "SortedMerList.h"

namespace mems {

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
    // uploads the sequences to the device and registers one SML per genome (mauveAligner.cpp:456); the entries of
    // an SML are only brought to the host when something reads them (SortedMerList.h)
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
    // sequences -> resident 2-bit genomes; the contig starts of multi-record sequences and the positions of
    // ambiguous bases go along (mauve_set_genomes_contigs): no seed, match or gapped alignment runs across them
    void upload(HipContext &hc) const
    {
        std::vector<std::vector<uint64_t>> packed(seq_table.size());
        std::vector<const uint64_t *> ptr; std::vector<int64_t> lens;
        std::vector<int64_t> n_contigs, contig_starts;
        std::vector<std::vector<uint64_t>> invalid(seq_table.size());
        std::vector<const uint64_t *> inv_ptr;
        for (size_t i = 0; i < seq_table.size(); i++) {
            const std::string &s = seq_table[i]->str();
            packed[i].assign(mauve_packed_words((int64_t)s.size()), 0);
            mauve_pack_ascii(s.data(), (int64_t)s.size(), packed[i].data());
            invalid[i].assign(((size_t)s.size() + 63) / 64 + 1, 0);
            mauve_ambiguity_bitmap(s.data(), (int64_t)s.size(), invalid[i].data());
            ptr.push_back(packed[i].data()); lens.push_back((int64_t)s.size()); inv_ptr.push_back(invalid[i].data());
            const std::vector<int64_t> &cs = seq_table[i]->contigStarts();
            n_contigs.push_back((int64_t)cs.size());
            contig_starts.insert(contig_starts.end(), cs.begin(), cs.end());
        }
        hc.check(mauve_set_genomes_contigs(hc.get(), (int)seq_table.size(), ptr.data(), lens.data(), n_contigs.data(), contig_starts.data(), inv_ptr.data()),
                 "mauve_set_genomes_contigs");
    }
    void MultiplicityFilter(uint mult)                                     // mauveAligner.cpp:600
    {
        size_t k = 0;
        for (size_t i = 0; i < size(); i++) { if ((*this)[i]->Multiplicity() == mult) (*this)[k++] = (*this)[i]; else (*this)[i]->Free(); }
        resize(k);
    }
    void Clear() { for (Match *m : *this) m->Free(); clear(); }            // MLDeleter, mauveAligner.cpp:39-45
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

// LoadSequences(list, log) (mauveAligner.cpp:454,463; addUnalignedIntervals.cpp:23; getAlignmentWindows.cpp:58 ...): one
// gnSequence per entry of seq_filename into seq_table (FastA; a multi-record file is one genome of several contigs).
// Works on anything with seq_filename / seq_table (MatchList, IntervalList).
template <class ListT>
inline void LoadSequences(ListT &list, std::ostream *log)
{
    for (size_t i = 0; i < list.seq_table.size(); i++) delete list.seq_table[i];
    list.seq_table.clear();
    for (size_t i = 0; i < list.seq_filename.size(); i++) {
        genome::gnSequence *s = new genome::gnSequence();
        s->LoadSource(list.seq_filename[i]);
        list.seq_table.push_back(s);
        if (log) *log << "Sequence loaded successfully.\n" << list.seq_filename[i] << " " << s->length() << " base pairs.\n";
    }
}

// LoadMFASequences(list, mfa_file, log) (alignmentProjector.cpp:55, evd.cpp:97): every record of ONE multi-FastA file
// becomes a genome of its own; seq_filename gets the file name once per record.
template <class ListT>
inline void LoadMFASequences(ListT &list, const std::string &mfa_filename, std::ostream *log)
{
    std::ifstream in(mfa_filename.c_str());
    if (!in) throw genome::gnException("LoadMFASequences: cannot open " + mfa_filename);
    for (size_t i = 0; i < list.seq_table.size(); i++) delete list.seq_table[i];
    list.seq_table.clear(); list.seq_filename.clear();
    std::string line, cur; bool have = false;
    auto flush = [&]() {
        if (!have) return;
        list.seq_table.push_back(new genome::gnSequence(cur)); list.seq_filename.push_back(mfa_filename);
        if (log) *log << "Sequence loaded successfully.\n" << mfa_filename << " " << cur.size() << " base pairs.\n";
        cur.clear();
    };
    while (std::getline(in, line)) {
        if (!line.empty() && line[0] == '>') { flush(); have = true; continue; }
        if (!have) continue;
        for (char ch : line) if (ch != '\r' && ch != ' ') cur.push_back(ch);
    }
    flush();
}

}  // namespace mems
#endif
