// libMems/PairwiseScoringScheme.h -- substitution matrix and affine gap penalties of the gapped aligner
// (progressiveMauve.cpp:666-687 `PairwiseScoringScheme(matrix, gap_open, gap_extend)`, readSubstitutionMatrix :684;
// repeatoire.cpp:1994 `PairwiseScoringScheme(hoxd_matrix, -100, -20)`).  Defaults: HOXD70, -400 / -30.
#ifndef MAUVE_HIP_PAIRWISESCORINGSCHEME_H
#define MAUVE_HIP_PAIRWISESCORINGSCHEME_H

#include <istream>
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

}  // namespace mems
#endif
