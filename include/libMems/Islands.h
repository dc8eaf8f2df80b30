// libMems/Islands.h -- the island / backbone outputs of mauveAligner (mauveAligner.cpp:807-847) and of the in-tree
// tools (extractBackbone.cpp:65, calculateBackboneCoverage.cpp:96): simpleFindBackbone, outputBackbone,
// simpleFindIslands, findIslandsBetweenLCBs.  All four are views of the device stage behind detectBackbone
// (Backbone.h):
//   backbone (all sequences)  = the segments whose genome set is every sequence, found with island gap = max_gap_size,
//                               kept when every sequence has at least backbone_size bases in them;
//   island (pairwise)         = a run of at least island_size columns in which one sequence of a pair has bases and the
//                               other has none;
//   island between LCBs       = a stretch of at least island_size bases of one sequence between two consecutive aligned
//                               intervals (or before the first one).
// Text layouts (libMems' own are not in the reference tree): backbone rows "left<TAB>right" per sequence, signed;
// island rows  "seq<TAB>left<TAB>right<TAB>other_seq<TAB>interval";  between-LCB rows  "seq<TAB>left<TAB>right".
#ifndef MAUVE_HIP_ISLANDS_H
#define MAUVE_HIP_ISLANDS_H

#include "Backbone.h"
#include "GappedAlignment.h"

namespace mems {

inline void simpleFindBackbone(IntervalList &il, uint backbone_size, uint max_gap_size, std::vector<GappedAlignment> &backbone_data)
{
    backbone_data.clear();
    backbone_list_t bb;
    detectBackboneAndIslands(il, max_gap_size, bb, nullptr);
    const uint N = (uint)il.seq_table.size();
    const uint32_t all = N >= 32 ? 0xffffffffu : (1u << N) - 1;
    std::vector<std::string> rows;
    for (size_t i = 0; i < bb.size(); i++) {
        bool have_rows = false;
        for (const BackboneSegment &b : bb[i]) {
            if (b.genomes != all) continue;
            bool long_enough = true;
            for (uint g = 0; g < N; g++) long_enough = long_enough && b.Length(g) >= backbone_size;
            if (!long_enough) continue;
            if (!have_rows) { il[i].GetAlignment(rows, il.seq_table); have_rows = true; }
            GappedAlignment ga(N, b.length);
            std::vector<std::string> cut(N);
            for (uint g = 0; g < N; g++) cut[g] = rows[g].substr((size_t)b.left_col, (size_t)b.length);
            ga.SetAlignment(cut);
            for (uint g = 0; g < N; g++) ga.SetStart(g, b.Start(g));
            backbone_data.push_back(ga);
        }
    }
}

inline void outputBackbone(const std::vector<GappedAlignment> &backbone_data, std::ostream &os)
{
    for (const GappedAlignment &ga : backbone_data) {
        for (uint g = 0; g < ga.SeqCount(); g++) {
            const int64 s = ga.Start(g), l = (int64)std::llabs(s), r = l + (int64)ga.Length(g) - 1;
            os << (g ? "\t" : "") << s << '\t' << (s < 0 ? -r : r);
        }
        os << '\n';
    }
}

inline void simpleFindIslands(IntervalList &il, uint island_size, std::ostream &os)
{
    backbone_list_t bb; std::vector<PairIsland> isl;
    detectBackboneAndIslands(il, island_size ? island_size - 1 : 0, bb, &isl);
    for (const PairIsland &p : isl)
        os << p.who << '\t' << p.left_end << '\t' << p.right_end << '\t' << (p.who == p.seq_a ? p.seq_b : p.seq_a) << '\t' << p.iv << '\n';
}

inline void findIslandsBetweenLCBs(IntervalList &il, uint island_size, std::ostream &os)
{
    const uint N = (uint)il.seq_table.size();
    for (uint g = 0; g < N; g++) {
        std::vector<std::pair<int64, int64>> cov;
        for (const Interval &iv : il) if (iv.Multiplicity() >= 2 && g < iv.SeqCount() && iv.LeftEnd(g)) cov.push_back(std::make_pair((int64)iv.LeftEnd(g), (int64)iv.RightEnd(g)));
        std::sort(cov.begin(), cov.end());
        int64 next = 1;
        for (const auto &cv : cov) {
            if (cv.first - next >= (int64)island_size && cv.first > next) os << g << '\t' << next << '\t' << cv.first - 1 << '\n';
            next = std::max(next, cv.second + 1);
        }
        const int64 len = g < il.seq_table.size() && il.seq_table[g] ? (int64)il.seq_table[g]->length() : 0;
        if (len >= next && len - next + 1 >= (int64)island_size) os << g << '\t' << next << '\t' << len << '\n';
    }
}

}  // namespace mems
#endif
