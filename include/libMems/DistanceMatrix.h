// libMems/DistanceMatrix.h -- the matrices of the call sites: DistanceMatrix(seq_count, coverage_list, distance) before
// MuscleInterface::CreateTree (mauveAligner.cpp:616-623) and IdentityMatrix(interval_list, identity) for --lcb-stats
// (mauveAligner.cpp:798-800; calculateBackboneCoverage.cpp:115).  Host code on small inputs; the guide tree of the
// alignment path itself is mauve_guide_tree (device).
//   coverage_list : (set of sequences as a bit mask, bases covered by matches of exactly that set)
//   distance(i,j) = 1 - shared(i,j) / min(total(i), total(j)),  shared = bases in sets holding both, total = holding the one
//   identity(i,j) = identical columns / columns where both have a base, over all intervals
#ifndef MAUVE_HIP_DISTANCEMATRIX_H
#define MAUVE_HIP_DISTANCEMATRIX_H
#include "IntervalList.h"
#include "NumericMatrix.h"
#include <cctype>
namespace mems {

inline void DistanceMatrix(uint seq_count, const std::vector<std::pair<uint64, uint64>> &coverage_list, NumericMatrix<double> &distance)
{
    distance.init(seq_count, seq_count);
    std::vector<double> total(seq_count, 0);
    NumericMatrix<double> shared(seq_count, seq_count);
    for (const auto &cv : coverage_list)
        for (uint i = 0; i < seq_count; i++) {
            if (!(cv.first >> i & 1)) continue;
            total[i] += (double)cv.second;
            for (uint j = i + 1; j < seq_count; j++) if (cv.first >> j & 1) { shared(i, j) += (double)cv.second; shared(j, i) += (double)cv.second; }
        }
    for (uint i = 0; i < seq_count; i++)
        for (uint j = 0; j < seq_count; j++) {
            const double mn = std::min(total[i], total[j]);
            distance(i, j) = i == j ? 0.0 : (mn > 0 ? 1.0 - std::min(1.0, shared(i, j) / mn) : 1.0);
        }
}

inline void IdentityMatrix(const IntervalList &iv_list, NumericMatrix<double> &identity)
{
    const uint N = (uint)iv_list.seq_table.size();
    identity.init(N, N);
    NumericMatrix<double> both(N, N);
    std::vector<std::string> rows;
    for (const Interval &iv : iv_list) {
        if (iv.Multiplicity() < 2) continue;
        iv.GetAlignment(rows, iv_list.seq_table);
        for (uint i = 0; i < N && i < rows.size(); i++)
            for (uint j = i + 1; j < N && j < rows.size(); j++) {
                double same = 0, shared = 0;
                for (size_t c = 0; c < rows[i].size(); c++) {
                    if (rows[i][c] == '-' || rows[j][c] == '-') continue;
                    shared += 1; same += toupper((unsigned char)rows[i][c]) == toupper((unsigned char)rows[j][c]);
                }
                identity(i, j) += same; identity(j, i) += same; both(i, j) += shared; both(j, i) += shared;
            }
    }
    for (uint i = 0; i < N; i++) for (uint j = 0; j < N; j++) identity(i, j) = i == j ? 1.0 : (both(i, j) > 0 ? identity(i, j) / both(i, j) : 0.0);
}

}  // namespace mems
#endif
