/*
 * mauve_oracle.h -- CPU restatement ("oracle") of the mauveAligner / progressiveMauve hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under mauvealigner_amd/ may include, link or call this.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY UNPINNED: the reference's arithmetic for this path lives in libMems 1.6 / libGenome /
 * libMUSCLE (configure.ac:46), which are absent from /root/reference and from this image, and the
 * reference ships no tests, fixtures or golden vectors.  This restatement
 * follows the in-tree call sites and subclasses (cited per function) and the published Mauve
 * algorithm; every semantic choice that the tree does not pin is frozen in DESIGN.md.
 */
#ifndef MAUVE_ORACLE_H
#define MAUVE_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_SEQ 32
#define ORC_CODING_SEED 3                 /* mauveAligner.cpp:266-279 seed ranks */
#define ORC_SOLID_SEED 0x7fffffff         /* repeatoire.cpp:1847: SOLID_SEED == INT_MAX */

/* seed-hit rule (who calls HashMatch) */
#define ORC_MODE_MEM 0        /* MemHash default: a mer repeated in any genome is dropped entirely */
#define ORC_MODE_UNIQUE 1     /* UniqueMatchFinder.cpp:36-60: drop only the genomes with repeats */
#define ORC_MODE_PAIRWISE 2   /* PairwiseMatchFinder (progressiveMauve.cpp:496-501): MemHash on each pair alone */

typedef struct {
    int32_t gap_open;         /* cost of the first gap column of a run (negative) */
    int32_t gap_extend;       /* cost of each further gap column (negative) */
    int32_t matrix[4][4];     /* substitution scores, A,C,G,T order */
} orc_scoring;

typedef struct {
    int64_t n;                /* number of matches */
    int32_t nseq;
    int64_t *length;          /* [n] */
    int64_t *start;           /* [n*nseq] signed 1-based, 0 = absent (NO_MATCH) */
} orc_matches;

typedef struct {
    int64_t n_lcb;
    int32_t nseq;
    int64_t *match_lcb;       /* [n_matches] LCB id or -1 (eliminated) */
    int64_t *left_end;        /* [n_lcb*nseq] signed (negative = reverse strand) */
    int64_t *right_end;       /* [n_lcb*nseq] signed */
    int64_t *weight;          /* [n_lcb] */
    int64_t *left_adj;        /* [n_lcb*nseq] LCB id left of this one in genome g, -1 = none */
    int64_t *right_adj;       /* [n_lcb*nseq] */
} orc_lcbs;

typedef struct {
    int64_t n_iv;             /* number of intervals (LCBs, then unaligned single-genome islands) */
    int32_t nseq;
    int64_t *left;            /* [n_iv*nseq] 1-based left end, 0 = genome absent */
    int64_t *right;           /* [n_iv*nseq] 1-based inclusive right end */
    int8_t  *reverse;         /* [n_iv*nseq] 1 = reverse strand */
    int64_t *col_off;         /* [n_iv+1] offsets into cols */
    uint32_t *cols;           /* presence mask per alignment column (bit g = genome g has a base) */
    int64_t *dp_score;        /* [n_iv] sum of DP scores of the gapped intervals of the LCB */
    int64_t n_anchor;         /* anchors used (after overlap trimming) */
    int64_t *anchor_length;   /* [n_anchor] */
    int64_t *anchor_start;    /* [n_anchor*nseq] signed 1-based */
    int64_t *anchor_lcb;      /* [n_anchor] */
    int64_t n_gap_dp;         /* number of inter-anchor intervals that went through DP */
    int64_t n_dp_cells;       /* total DP cells evaluated */
} orc_alignment;

typedef struct {
    uint64_t seed_pattern;    /* 0 -> getSeed(default weight, seed_rank) */
    int32_t seed_weight;      /* 0 -> default (mauveAligner.cpp:92) */
    int32_t seed_rank;
    int32_t mode;             /* ORC_MODE_* */
    int64_t lcb_weight;       /* -1 -> 3*weight*N (mauveAligner.cpp:648-653); already multiplied by N */
    int32_t collinear;        /* mauveAligner.cpp:665-666 */
    int32_t recursive;        /* mauveAligner.cpp:94 */
    int32_t gapped;           /* mauveAligner.cpp:96 */
    int32_t add_unaligned;    /* mauveAligner.cpp:748 addUnalignedIntervals */
    int32_t extend_lcbs;      /* lcb_extension (mauveAligner.cpp:95); frozen replacement: DESIGN.md S10 */
    int32_t max_extension_iters; /* default 4 (mauveAligner.cpp:687-690) */
    int64_t min_recursive_gap;/* default 200 (mauveAligner.cpp:899) */
    int64_t max_gapped_len;   /* default 10000 */
    orc_scoring scoring;
    int64_t max_banded_len;   /* banded DP for intervals in (max_gapped_len, max_banded_len]; default 0 = off */
    int32_t lcb_scoring;      /* 0 = length weights (Aligner::align), 1 = extant sum-of-pairs anchor scores */
    int32_t weight_scaling;   /* node weight x (1 - conservation_scale x conservation distance of the node) */
    int32_t conservation_scale_ppm;
    int32_t seed_family;      /* search with the three seeds of the weight, longest first, contained matches dropped */
    int64_t min_scaled_penalty;
    int32_t refine_rounds;    /* every gapped interval of >= 3 sequences is also aligned in up to this many rotated orders, the best sum-of-pairs score is kept; 0 = off */
    int32_t bp_dist_scale_ppm;/* node weight x (1 - scale x breakpoint distance of the node); with weight_scaling only */
    int64_t bp_dist_min_score;/* pairwise matches shorter than this do not count towards the breakpoint estimate; -1 = 2 x seed weight */
} orc_params;

/* ---- seeds ---------------------------------------------------------------------------------- */
uint64_t orc_get_seed(int weight, int rank);
int orc_seed_length(uint64_t pattern);
int orc_seed_weight(uint64_t pattern);
int orc_default_seed_weight(int64_t avg_len);
void orc_default_scoring(orc_scoring *s);
void orc_default_params(orc_params *p);
void orc_default_progressive_params(orc_params *p);

/* ---- sequence encoding ------------------------------------------------------------------------ */
void orc_encode(const char *ascii, int64_t n, uint8_t *codes);       /* A,C,G,T -> 0..3, other -> 0 */
void orc_pack2bit(const uint8_t *codes, int64_t n, uint32_t *words); /* base i -> word i/16 bits 2*(i%16) */

/* ---- sorted mer list (SML) -------------------------------------------------------------------- */
/* per-position canonical masked mer + strand flag; returns number of positions (L-span+1 or 0) */
int64_t orc_mers(const uint8_t *codes, int64_t len, uint64_t pattern, uint64_t *canon, uint8_t *strand);
/* SML: positions sorted by (canon mer, position); mer_out is libMems-style: left-aligned | strand */
int64_t orc_sorted_mer_list(const uint8_t *codes, int64_t len, uint64_t pattern,
                            uint64_t *mer_out, int64_t *pos_out);

/* ---- multi-MUM enumeration + ungapped extension ---------------------------------------------- */
int orc_find_matches(int nseq, const uint8_t *const *codes, const int64_t *lens, uint64_t pattern,
                     int mode, uint64_t mask, int extend, orc_matches *out);
/* union of two match lists for a seed family: a, then the matches of b no match of a contains; canonical order */
int orc_merge_matches(int nseq, const orc_matches *a, const orc_matches *b, orc_matches *out);
/* the same search restricted to the bases inside the given intervals (1-based inclusive, sorted, disjoint,
   CSR by genome: iv_off[nseq+1]); a window is valid only if it holds no base outside them */
int orc_find_matches_masked(int nseq, const uint8_t *const *codes, const int64_t *lens, uint64_t pattern,
                            int mode, uint64_t mask, int extend, const int64_t *iv_off, const int64_t *iv_lo,
                            const int64_t *iv_hi, orc_matches *out);
/* SeedMatchEnumerator (SeedMatchEnumerator.h:19-141): single genome, every repeated seed -> Match */
int orc_seed_match_enumerate(const uint8_t *codes, int64_t len, uint64_t pattern, int64_t min_multi,
                             int64_t max_multi, int direct_only, int64_t *n_out, int64_t **mult_out,
                             int64_t **start_off_out, int64_t **starts_out);
void orc_free_matches(orc_matches *m);
void orc_free(void *p);

/* ---- LCBs ------------------------------------------------------------------------------------ */
int orc_multiplicity_filter(const orc_matches *in, int mult, orc_matches *out);
int orc_eliminate_overlaps(orc_matches *m);   /* in place, N-way input */
int orc_compute_lcbs_w(const orc_matches *m, const int64_t *match_weight, int64_t min_weight, int collinear, orc_lcbs *out);
void orc_match_sp_scores(int nseq, const uint8_t *const *codes, const orc_matches *m, const orc_scoring *sc, int64_t *out);
int orc_compute_lcbs(const orc_matches *m, int64_t min_weight, int collinear, orc_lcbs *out);
void orc_free_lcbs(orc_lcbs *l);

/* ---- gapped DP -------------------------------------------------------------------------------- */
/* progressive N-way alignment of one inter-anchor interval; seqs[g] are codes already in LCB
   orientation; cols_out must hold sum(lens) entries; returns number of columns, score in *score */
/* banded variants: cells outside |j - floor(i*n/m)| <= 128 + |n-m| + (m+n)/64 are minus infinity */
int64_t orc_profile_dp_band(int64_t m, const uint8_t *cnt, int k_rows, int64_t n, const uint8_t *seq,
                            const orc_scoring *sc, uint8_t *ops_out, int64_t *score, int banded);
int64_t orc_align_interval_band(int nseq, const uint8_t *const *seqs, const int64_t *lens,
                                const orc_scoring *sc, uint32_t *cols_out, int64_t *score, int64_t *cells, int banded);
int64_t orc_band_cells(int64_t m, int64_t n, int banded);
int64_t orc_align_interval(int nseq, const uint8_t *const *seqs, const int64_t *lens,
                           const orc_scoring *sc, uint32_t *cols_out, int64_t *score,
                           int64_t *cells);
/* one profile-vs-sequence step (exposed for known-answer tests): profile given as per-column
   base counts cnt[m*4] and residue count; ops_out[m+n]: 1=profile only,2=seq only,3=both */
int64_t orc_profile_dp(int64_t m, const uint8_t *cnt, int k_rows, int64_t n, const uint8_t *seq,
                       const orc_scoring *sc, uint8_t *ops_out, int64_t *score);

/* ---- whole path -------------------------------------------------------------------------------- */
int orc_align(int nseq, const uint8_t *const *codes, const int64_t *lens, const orc_params *p,
              orc_matches *mums_out, orc_lcbs *lcbs_out, orc_alignment *aln_out);
void orc_free_alignment(orc_alignment *a);
/* guide tree + guide-tree recursive anchoring (ProgressiveAligner::align stand-in);
   dist: [nseq*nseq] ppm distances (may be NULL), tree_left/right: [2*nseq-1] */
/* sum-of-pairs score of the columns of one interval (substitution scores of both-columns, affine gap runs per pair) */
int64_t orc_sp_score_cols(int nseq, const uint8_t *const *seqs, const int64_t *lens, const uint32_t *cols, int64_t ncols, const orc_scoring *sc);
/* orc_align_interval + the rotated orders; cells counts every DP that was run */
int64_t orc_align_interval_refined(int nseq, const uint8_t *const *seqs, const int64_t *lens, const orc_scoring *sc, int rounds,
                                   uint32_t *cols_out, int64_t *score, int64_t *cells);
/* broken adjacencies between the pairwise matches (length >= min_len) of every genome pair; bp[nseq*nseq], symmetric */
int orc_breakpoint_counts(int nseq, const uint8_t *const *codes, const int64_t *lens, uint64_t pattern, int64_t min_len, int64_t *bp);
int orc_guide_tree(int nseq, const uint8_t *const *codes, const int64_t *lens, uint64_t pattern,
                   int64_t *dist, int32_t *left, int32_t *right);
int orc_progressive_align(int nseq, const uint8_t *const *codes, const int64_t *lens, const orc_params *p,
                          int32_t *tree_left, int32_t *tree_right, int64_t *dist, orc_alignment *aln);
int orc_check_tree(int nseq, const int32_t *tree_left, const int32_t *tree_right);
int orc_progressive_align_tree(int nseq, const uint8_t *const *codes, const int64_t *lens, const orc_params *p,
                               const int32_t *tree_left, const int32_t *tree_right, orc_alignment *aln);
/* backbone segments and pairwise islands of an alignment (stands in for libMems detectBackbone with
   BigGapsDetector, progressiveMauve.cpp:242-243, and simpleFindIslands, mauveAligner.cpp:844) */
typedef struct {
    int32_t nseq;
    int64_t n_seg;
    int64_t *seg_iv, *seg_col, *seg_len;  /* [n_seg] interval, first column in it, columns */
    uint32_t *seg_mask;                   /* [n_seg] genomes with residues in the segment (>= 2) */
    int64_t *seg_left, *seg_right;        /* [n_seg*nseq] signed ends (negative = reverse), 0 = not in the segment */
    int64_t n_isl;
    int64_t *isl;                         /* [n_isl*8] interval, a, b (a < b), who has the residues, first col, last col, signed left, right */
} orc_backbone;
/* the homology pass in front of the backbone (detectAndApplyBackbone's HMM, progressiveMauve.cpp:226-243 [EXT]) */
typedef struct { int32_t match, mismatch, gap, go_homologous, go_unrelated; } orc_hmm_params;      /* log-odds and log transition probabilities x 1000 */
void orc_hmm_params_from(double identity, double pgh, double pgu, orc_hmm_params *h);
/* columns in, columns out (cols_out holds up to the number of residues; col_off_out [n_iv+1]); returns the residues taken out of their columns, < 0 on error */
int64_t orc_homology_apply(int nseq, const uint8_t *const *codes, int64_t n_iv, const int64_t *left, const int64_t *right, const int8_t *reverse,
                           const int64_t *col_off, const uint32_t *cols, const orc_hmm_params *h, int64_t *col_off_out, uint32_t *cols_out);
int orc_backbone_detect(int nseq, int64_t n_iv, const int64_t *left, const int64_t *right, const int8_t *reverse,
                        const int64_t *col_off, const uint32_t *cols, int64_t island_gap, orc_backbone *out);
void orc_free_backbone(orc_backbone *o);
/* XMFA text (format pinned by mfa2xmfa.cpp:64,89-91,104-115); returns malloc'd NUL-terminated text */
char *orc_write_xmfa(int nseq, const uint8_t *const *codes, const int64_t *lens,
                     const char *const *names, const orc_alignment *a, int64_t *text_len);

#ifdef __cplusplus
}
#endif
#endif
