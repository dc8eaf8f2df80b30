// libMems/PairwiseScoringScheme.h -- substitution matrix and affine gap penalties of the gapped aligner
// (progressiveMauve.cpp:666-687 `PairwiseScoringScheme(matrix, gap_open, gap_extend)`, readSubstitutionMatrix :684;
// repeatoire.cpp:1994 `PairwiseScoringScheme(hoxd_matrix, -100, -20)`).  Defaults: HOXD70, -400 / -30.
#ifndef MAUVE_HIP_PAIRWISESCORINGSCHEME_H
#define MAUVE_HIP_PAIRWISESCORINGSCHEME_H

#include <istream>
#include <string>
#include <vector>
#include "AbstractMatch.h"

namespace mems {

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


// HOXD70 as libMems exports it (repeatoire.cpp:1994)
static const score_t hoxd_matrix[4][4] = {{91, -114, -31, -123}, {-114, 100, -125, -31}, {-31, -125, 100, -114}, {-123, -31, -114, 91}};

// progressiveMauve.cpp:684: a 4 x 4 matrix in A, C, G, T order; lines starting with '#' and letter labels are skipped
inline void readSubstitutionMatrix(std::istream &is, score_t matrix[4][4])
{
    int got = 0; std::string tok;
    while (got < 16 && is >> tok) {
        if (tok[0] == '#') { std::string rest; std::getline(is, rest); continue; }
        char *end = nullptr; const long v = strtol(tok.c_str(), &end, 10);
        if (end == tok.c_str() || *end) continue;                 // row / column labels
        matrix[got / 4][got % 4] = (score_t)v; got++;
    }
    if (got != 16) throw genome::gnException("readSubstitutionMatrix: expected 16 scores");
}

// computeSPScore (repeatoire.cpp:2527: `computeSPScore(alignment, pss, scores_final, score_final)`): sum-of-pairs score
// of a gapped alignment given as one text row per sequence ('-' = gap).  libMems-internal [EXT]; frozen form, the pair
// scoring of DESIGN.md S7: for every pair of rows, a column with two bases scores matrix[a][b] (bases outside ACGT
// score as A, S1), a base against a gap opens (gap_open) or continues (gap_extend) a gap run of that pair, two gaps
// score nothing and do not interrupt a run.  scores[c] = the column's sum over the pairs, score = their total.
inline void computeSPScore(const std::vector<std::string> &alignment, const PairwiseScoringScheme &pss,
                           std::vector<score_t> &scores, score_t &score)
{
    const size_t R = alignment.size(), C = R ? alignment[0].size() : 0;
    for (size_t r = 1; r < R; r++) if (alignment[r].size() != C) throw genome::gnException("computeSPScore: ragged alignment");
    auto code = [](char ch) { switch (ch) { case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 0; } };
    scores.assign(C, 0); score = 0;
    for (size_t x = 0; x < R; x++)
        for (size_t y = x + 1; y < R; y++) {
            int open = 0;                                  // 1: a run with row x gapped is open, 2: row y gapped
            for (size_t c = 0; c < C; c++) {
                const bool gx = alignment[x][c] == '-', gy = alignment[y][c] == '-';
                if (gx && gy) continue;
                score_t v;
                if (!gx && !gy) { v = pss.matrix[code(alignment[x][c])][code(alignment[y][c])]; open = 0; }
                else { const int side = gx ? 1 : 2; v = open == side ? pss.gap_extend : pss.gap_open; open = side; }
                scores[c] += v; score += v;
            }
        }
}

}  // namespace mems
#endif
