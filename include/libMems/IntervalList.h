// libMems/IntervalList.h -- the result of Aligner::align: vector<Interval> plus seq_table / seq_filename, with the
// writers and readers at the stage seams: XMFA (WriteStandardAlignment, mauveAligner.cpp:746-760, layout pinned by
// mfa2xmfa.cpp:64-115; ReadStandardAlignment, scoreProcrastAlignment.cpp:442) and the interval list file.
#ifndef MAUVE_HIP_INTERVALLIST_H
#define MAUVE_HIP_INTERVALLIST_H

#include <sstream>
#include "Interval.h"
#include "SortedMerList.h"
#include "MatchList.h"                  // LoadSequences(IntervalList&, ostream*) is reached through this header (backbone_global_to_local.cpp:21)
#include <set>
#include <map>
#include <cmath>

namespace mems {

class IntervalList : public std::vector<Interval> {
public:
    std::vector<genome::gnSequence *> seq_table;
    std::vector<std::string> seq_filename;
    std::vector<std::string> defline_name;               // per sequence: the name an XMFA that was read carried in its deflines (else the file name is written)
    std::string backbone_filename;                       // progressiveMauve.cpp:259: the .bbcols file that goes with this list
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
                   << ' ' << defname(g) << '\n';
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
        clear(); seq_filename.clear(); defline_name.clear();
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
                if (g > MAUVE_MAX_SEQ) throw genome::gnException("ReadStandardAlignment: sequence index beyond the 32 this library aligns: " + line);
                r.g = g - 1; r.lo = lo; r.hi = hi; r.rev = strand == '-';
                {   // the name behind the strand sign: kept, so that what is read is written back as it was (mfa2xmfa.cpp:104-105 puts the record name there)
                    size_t p = line.find(' ', 2); if (p != std::string::npos) p = line.find(' ', p + 1);
                    if (defline_name.size() < g) defline_name.resize(g);
                    if (p != std::string::npos && defline_name[g - 1].empty()) defline_name[g - 1] = line.substr(p + 1);
                }
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

    // Interval list file (mauveAligner.cpp:702,715 write / read libMems' .mln, whose body layout is not reproduced
    // here).  This is a text layout of this library's own, tagged as such in its first line so that neither side
    // mistakes the other's files: header, then per interval the signed starts, lengths and the gap pattern
    // run-length encoded per genome; ReadList rejects anything without the tag.
    void WriteList(std::ostream &os) const
    {
        const uint N = (uint)seq_table.size();
        os << "FormatVersion\tmauve_hip_mln_1\nSequenceCount\t" << N << '\n';
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
        expect("FormatVersion"); std::string ver; is >> ver;
        if (ver != "mauve_hip_mln_1") throw genome::gnException("IntervalList::ReadList: not an interval list written by this library (libMems' own .mln layout is not read): FormatVersion " + ver);
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
    std::string defname(uint g) const { return g < defline_name.size() && !defline_name[g].empty() ? defline_name[g] : name(g); }
};


// mauveAligner.cpp:748: every stretch of every sequence that no interval covers becomes a single-sequence interval
inline void addUnalignedIntervals(IntervalList &il)
{
    const uint N = (uint)il.seq_table.size();
    std::vector<Interval> extra;
    for (uint g = 0; g < N; g++) {
        std::vector<std::pair<int64, int64>> sp;
        for (const Interval &iv : il) if (g < iv.SeqCount() && iv.LeftEnd(g)) sp.push_back(std::make_pair((int64)iv.LeftEnd(g), (int64)iv.RightEnd(g)));
        std::sort(sp.begin(), sp.end());
        int64 cur = 1;
        for (size_t i = 0; i <= sp.size(); i++) {
            const int64 lo = cur, hi = i < sp.size() ? sp[i].first - 1 : (int64)il.seq_table[g]->length();
            if (hi >= lo) {
                std::vector<int64> l(N, 0), r(N, 0); std::vector<char> rv(N, 0);
                l[g] = lo; r[g] = hi;
                extra.push_back(Interval(l, r, rv, std::vector<uint32_t>((size_t)(hi - lo + 1), 1u << g)));
            }
            if (i < sp.size() && sp[i].second + 1 > cur) cur = sp[i].second + 1;
        }
    }
    il.insert(il.end(), extra.begin(), extra.end());
}

}  // namespace mems
#endif
