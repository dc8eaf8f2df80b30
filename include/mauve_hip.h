/*
 * mauve_hip.h -- C-ABI of the MI355X-native mauveAligner hot path (libmauve_hip.so).
 *
 * This is the drop-in boundary.  The reference has no C plugin ABI: its seams are the
 * C++ virtual interfaces of libMems.  Each entry point below names the libMems interface it stands
 * behind, cited by the in-tree call site that pins its contract (file:line relative to the
 * reference tree).  The C++ adapters in include/libMems/ (namespace mems) call these and nothing
 * else; INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions: plain pointers and sizes, caller-owned output buffers (two-phase count / fill),
 * every call returns an int status (0 = MAUVE_OK) and never throws; mauve_last_error() gives the
 * text.  A context is thread-compatible (one thread at a time); there are no hidden globals.
 * Coordinates follow libMems: signed 1-based starts, negative = reverse strand, 0 = NO_MATCH
 * (SeedMatchEnumerator.h:83,127-141; sortContigs.cpp:46).
 */
#ifndef MAUVE_HIP_H
#define MAUVE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAUVE_OK 0
#define MAUVE_ERR_ARG (-1)        /* bad argument */
#define MAUVE_ERR_HIP (-2)        /* HIP runtime failure (text in mauve_last_error) */
#define MAUVE_ERR_NOGPU (-3)      /* no usable device: the product has no CPU fallback */
#define MAUVE_ERR_LIMIT (-4)      /* input exceeds a documented limit */
#define MAUVE_ERR_STATE (-5)      /* call order violated (e.g. no genomes set) */

#define MAUVE_MAX_SEQ 32
#define MAUVE_MAX_SEED_SPAN 49
#define MAUVE_CODING_SEED 3                /* mauveAligner.cpp:266-279 */
#define MAUVE_SOLID_SEED 0x7fffffff        /* repeatoire.cpp:1847 (SOLID_SEED == INT_MAX) */

/* seed-hit rule = which MatchFinder subclass's EnumerateMatches is in force */
#define MAUVE_MODE_MEM 0      /* mems::MemHash / MaskedMemHash (mauveAligner.cpp:523-531) */
#define MAUVE_MODE_UNIQUE 1   /* UniqueMatchFinder::EnumerateMatches (UniqueMatchFinder.cpp:36-60) */
#define MAUVE_MODE_PAIRWISE 2 /* mems::PairwiseMatchFinder (progressiveMauve.cpp:496-501): MemHash on every genome
                                pair separately; matches have exactly two components; mask is ignored */

#define MAUVE_LCB_SCORE_LENGTH 0
#define MAUVE_LCB_SCORE_SP 1

typedef struct mauve_ctx mauve_ctx;

typedef struct {
    int32_t gap_open;             /* PairwiseScoringScheme.gap_open  (progressiveMauve.cpp:666-687) */
    int32_t gap_extend;           /* PairwiseScoringScheme.gap_extend */
    int32_t matrix[4][4];         /* score_t matrix[4][4], A,C,G,T (readSubstitutionMatrix, :684) */
} mauve_scoring;

typedef struct {
    uint64_t seed_pattern;        /* 0 -> getSeed(weight, rank) */
    int32_t seed_weight;          /* 0 -> getDefaultSeedWeight(avg length)  (mauveAligner.cpp:92,650) */
    int32_t seed_rank;            /* mauveAligner.cpp:93 */
    int32_t mode;                 /* MAUVE_MODE_* */
    int64_t lcb_weight;           /* -1 -> 3*weight*N (mauveAligner.cpp:648-653); total, i.e. already *N */
    int32_t collinear;            /* mauveAligner.cpp:665-666 */
    int32_t recursive;            /* mauveAligner.cpp:94 */
    int32_t gapped;               /* mauveAligner.cpp:96 */
    int32_t add_unaligned;        /* mauveAligner.cpp:748 addUnalignedIntervals */
    int32_t extend_lcbs;          /* lcb_extension, mauveAligner.cpp:95: default 1, as there (frozen form DESIGN.md S10; not applied to score-weighted LCBs) */
    int32_t max_extension_iters;  /* Aligner::SetMaxExtensionIterations, default 4 (mauveAligner.cpp:687-690) */
    int64_t min_recursive_gap;    /* Aligner::SetMinRecursionGapLength, default 200 (:670-672,899) */
    int64_t max_gapped_len;       /* Aligner::SetMaxGappedAlignmentLength (:674-676), default 10000 */
    mauve_scoring scoring;
    int64_t max_banded_len;       /* no reference counterpart (its aligner leaves intervals above max_gapped_len unaligned): intervals
                                     whose longest sequence is in (max_gapped_len, max_banded_len] are aligned by the banded DP
                                     (DESIGN.md S7b); default 0 = off */
    int32_t lcb_scoring;          /* ProgressiveAligner::setLcbScoringScheme (progressiveMauve.cpp:611-625): MAUVE_LCB_SCORE_LENGTH
                                     (0, default: weight = sum of length * n, the Aligner::align rule) or MAUVE_LCB_SCORE_SP (1: extant
                                     sum-of-pairs score of the anchors); lcb_weight is then a score */
    int32_t weight_scaling;   /* progressive path: scale every node's minimum LCB weight by the conservation distance of its two
                                 subtrees (ProgressiveAligner::setUseLcbWeightScaling, progressiveMauve.cpp:626-627); default 0 */
    int32_t conservation_scale_ppm;   /* setConservationDistanceScale in parts per million (:633-637; call-site default 0.5 = 500000) */
    int32_t seed_family;      /* search with the family of three seeds of the weight instead of one (progressiveMauve.cpp:502-546 --seed-family;
                                 ProgressiveAligner::setUseSeedFamilies :604-605); default 0 */
    int64_t min_scaled_penalty;       /* setMinimumBreakpointPenalty (:649-652): floor of the scaled weight; default 0 */
    int32_t refine_rounds;    /* progressive path, ProgressiveAligner::setRefinement (progressiveMauve.cpp:578-579): every gapped interval of >= 3
                                 sequences is also aligned in up to this many rotated orders and the alignment with the best sum-of-pairs score
                                 is kept (frozen form DESIGN.md S13); default 0 = off, the mirror's ProgressiveAligner starts with 2 */
    int32_t bp_dist_scale_ppm;/* setBreakpointDistanceScale in parts per million (--max-breakpoint-distance-scale, :628-632; call-site default
                                 0.5): with weight_scaling, a node's minimum weight also shrinks with the breakpoint distance between its two
                                 subtrees (frozen form DESIGN.md S11c); default 0 */
    int64_t bp_dist_min_score;/* setBpDistEstimateMinScore (:638-642): pairwise matches shorter than this do not count towards the breakpoint
                                 estimate; -1 (default) = 2 x seed weight */
} mauve_params;

/* sizes of the result of mauve_align(), for the caller to allocate the fill buffers */
typedef struct {
    int64_t n_mums;               /* N-way multi-MUMs found by the seed pass */
    int64_t n_lcb;
    int64_t n_anchor;             /* anchors after overlap elimination + recursive anchoring */
    int64_t n_iv;                 /* intervals: LCBs first, then unaligned single-genome islands */
    int64_t n_cols;               /* total alignment columns over all intervals */
    int64_t n_gap_dp;             /* inter-anchor intervals aligned by DP */
    int64_t n_dp_cells;           /* DP cells evaluated */
} mauve_align_sizes;

/* ---- context ----------------------------------------------------------------------------------- */
int mauve_ctx_create(int device, mauve_ctx **out);
void mauve_ctx_destroy(mauve_ctx *ctx);
const char *mauve_last_error(const mauve_ctx *ctx);       /* ctx may be NULL: last create error */
int mauve_device_name(const mauve_ctx *ctx, char *buf, size_t buflen);
int mauve_synchronize(mauve_ctx *ctx);
/* Page-locked host memory for the caller's bulk buffers (no reference counterpart: libMems keeps everything in pageable
   std::vectors).  Optional: every entry point takes any host pointer.  mauve_set_genomes uploads packed genomes that
   live in such a buffer straight from it, and mauve_align_fetch copies the column array straight into one (one DMA,
   no staging copy); pageable buffers go through the context's page-locked staging as before. */
int mauve_host_alloc(size_t bytes, void **out);
void mauve_host_free(void *p);

/* ---- seeds (host-side helpers; libMems free functions getSeed/getSeedLength/getDefaultSeedWeight,
        progressiveMauve.cpp:217,511-517; MatchList::GetDefaultMerSize, mauveAligner.cpp:651) ------ */
uint64_t mauve_get_seed(int weight, int rank);
int mauve_seed_length(uint64_t pattern);
int mauve_seed_weight(uint64_t pattern);
int mauve_default_seed_weight(int64_t avg_len);
void mauve_default_scoring(mauve_scoring *s);              /* hoxd_matrix, -400, -30 */
void mauve_default_params(mauve_params *p);                /* mauveAligner's call site (mauveAligner.cpp:92-99) */
/* progressiveMauve's call site: what a ProgressiveAligner does when main() sets nothing -- ExtantSumOfPairsScoring
   (progressiveMauve.cpp:624-625), LCB weight scaling on with conservation and breakpoint distance scales 0.5 (:285-287,626-637),
   refinement on (:578-579): lcb_scoring = SP, weight_scaling = 1, both *_scale_ppm = 500000, refine_rounds = 2. */
void mauve_default_progressive_params(mauve_params *p);
/* 2-bit packing used at the boundary: base i -> 64-bit word i/32, bits 2*(i%32); A,C,G,T=0..3,
   anything else -> 0.  words must hold mauve_packed_words(len) entries. */
size_t mauve_packed_words(int64_t len);
void mauve_pack_ascii(const char *ascii, int64_t len, uint64_t *words);
void mauve_pack_codes(const uint8_t *codes, int64_t len, uint64_t *words);

/* ---- genomes: gnSequence table of a MatchList (MatchList.seq_table; mauveAligner.cpp:453-465).
        Uploads the packed genomes; they stay resident in HBM until the next call / destroy. ------- */
int mauve_set_genomes(mauve_ctx *ctx, int nseq, const uint64_t *const *packed, const int64_t *lens);

/* The same for sequences that are concatenations of contigs and / or hold ambiguous bases (a multi-record FastA
   loaded as one gnSequence, mauveAligner.cpp:453-465; the contig-start table RepeatHashCat keeps for concatenated
   input, RepeatHashCat.h:19-20 `concat_contig_start`).  n_contigs[g] 0-based ascending starts per genome (the first
   is 0), concatenated in contig_starts; invalid[g]: bitmap of the bases that are not A, C, G, T (bit i of word i/64;
   mauve_ambiguity_bitmap builds it; the array or any entry may be NULL).  No seed window touches an ambiguous base
   or runs across a contig join: matches never cover an ambiguous base, and a match continues over a join only where
   windows on its two sides abut on one diagonal in every genome.  The packed genomes carry A at ambiguous bases
   (the gapped alignment sees A); mauve_write_xmfa prints N there. */
int mauve_set_genomes_contigs(mauve_ctx *ctx, int nseq, const uint64_t *const *packed, const int64_t *lens,
                              const int64_t *n_contigs, const int64_t *contig_starts, const uint64_t *const *invalid);
void mauve_ambiguity_bitmap(const char *ascii, int64_t len, uint64_t *bits);      /* bits: len/64 + 1 words */

/* ---- sorted mer list: MatchList::CreateMemorySMLs / DNAFileSML for one genome
        (mauveAligner.cpp:456,465; SortedMerList::GetMer semantics SeedMatchEnumerator.h:133).
        mer_out: mer left-aligned in 64 bits | strand flag in bit 0; pos_out: 0-based positions;
        both must hold len-span+1 entries.  Returns the entry count in *n_out. ------------------- */
int mauve_sorted_mer_list(mauve_ctx *ctx, int seq, uint64_t pattern, uint64_t *mer_out,
                          int64_t *pos_out, int64_t *n_out);

/* ---- multi-MUMs: MatchFinder::FindMatches(MatchList&) = AddSequence* + CreateMatches + GetMatchList
        (progressiveMauve.cpp:490-501; mauveAligner.cpp:577-589).  mask != 0 keeps only matches
        whose component set equals mask (MaskedMemHash::SetMask, mauveAligner.cpp:525-531);
        extend = 0 returns seed-length matches (SeedMatchEnumerator.h:71-123).  The result stays on
        the device; *n_matches receives the count.  mauve_get_matches copies it out in canonical
        order: length[n], start[n*nseq]. ---------------------------------------------------------- */
int mauve_seed_mums(mauve_ctx *ctx, uint64_t pattern, int mode, uint64_t mask, int extend,
                    int64_t *n_matches);
int mauve_get_matches(mauve_ctx *ctx, int64_t *length, int64_t *start);
/* ---- MemHash::HashMatch for seed hits a host-side finder enumerated itself: the host callback path of a MatchFinder
        subclass that overrides EnumerateMatches(IdmerList&) and calls HashMatch per hit (UniqueMatchFinder.cpp:36-60;
        the virtuals UniqueMatchFinder.h:31, SeedMatchEnumerator.h:38-39).  Hit h has the genomes of mask[h]; genome g's
        window starts at 0-based base pos[h*nseq+g] and strand[h*nseq+g] is the strand flag of its stored mer (bit 0 of
        SortedMerList::GetMer, SeedMatchEnumerator.h:133).  The hits are extended (extend != 0) and put in canonical
        order exactly like the hits of mauve_seed_mums; fetch with mauve_get_matches. ----------------------------- */
int mauve_extend_hits(mauve_ctx *ctx, uint64_t pattern, int64_t n_hits, const uint32_t *mask, const int64_t *pos,
                      const uint8_t *strand, int extend, int64_t *n_matches);

/* ---- SeedMatchEnumerator::FindMatches (SeedMatchEnumerator.h:19-33): single genome `seq`, every
        mer with min_multi..max_multi occurrences becomes one match (CSR output).  Two-phase: call
        with starts == NULL to get *n_out and *n_starts, then again with buffers. ----------------- */
int mauve_seed_match_enumerate(mauve_ctx *ctx, int seq, uint64_t pattern, int64_t min_multi,
                               int64_t max_multi, int direct_only, int64_t *n_out, int64_t *n_starts,
                               int64_t *mult, int64_t *start_off, int64_t *starts);

/* ---- chaining: MultiplicityFilter + EliminateOverlaps + Aligner::align's LCB stage
        (IdentifyBreakpoints / ComputeLCBs_v2 / greedy breakpoint elimination /
        computeLCBAdjacencies_v2; mauveAligner.cpp:596,600,698; toGrimmFormat.cpp:51-79).
        Host-side (sequential by nature); inputs/outputs are host arrays. ------ */
int mauve_eliminate_overlaps(int nseq, int64_t *n_inout, int64_t *length, int64_t *start);
/* one finder fed by the searches of several seeds (progressiveMauve.cpp:502-546: umf.FindMatches per seed of the family,
   longest first, then umf.GetMatchList): list a, then the matches of b that no match of a contains (same components,
   strands and diagonal), in canonical order.  Two-phase like the other list calls: with len_out == NULL
   *n_out receives the size; with buffers, *n_out is their capacity on entry. */
int mauve_merge_matches(int nseq, int64_t n_a, const int64_t *len_a, const int64_t *start_a, int64_t n_b, const int64_t *len_b,
                        const int64_t *start_b, int64_t *n_out, int64_t *len_out, int64_t *start_out);
int mauve_lcb_chain(int nseq, int64_t n, const int64_t *length, const int64_t *start,
                    int64_t min_weight, int collinear, int64_t *match_lcb, int64_t *n_lcb_out,
                    int64_t *left_end, int64_t *right_end, int64_t *weight, int64_t *left_adj,
                    int64_t *right_adj);   /* per-LCB arrays sized for n entries * nseq */

/* ---- gapped DP: the GappedAligner seam (Aligner::SetGappedAligner, mauveAligner.cpp:674;
        MuscleInterface::Align, MatchRecord.h:311; CallMuscleFast, repeatoire.cpp:1262).
        Batch of n_iv intervals; interval i has nseq sequences given as 0..3 codes, concatenated in
        `codes` with offsets seq_off[i*nseq+g] .. seq_off[i*nseq+g+1].  Output: presence-mask
        columns (bit g = sequence g has a residue), col_off[n_iv+1], per-interval score.
        cols must hold seq_off[n_iv*nseq] entries. ------------------------------------------------ */
int mauve_dp_batch(mauve_ctx *ctx, int nseq, int64_t n_iv, const uint8_t *codes,
                   const int64_t *seq_off, const mauve_scoring *sc, uint32_t *cols,
                   int64_t *col_off, int64_t *score);
/* The same seam for long intervals (GappedAligner::SetMaxAlignmentLength, GappedAligner.h: the reference's aligners
   refuse what is longer; this one bands it): an interval whose longest sequence is above band_from runs every
   progressive step inside the band |j - floor(i*n/m)| <= 128 + |n-m| + (m+n)/64; the others in full. */
int mauve_dp_batch_banded(mauve_ctx *ctx, int nseq, int64_t n_iv, const uint8_t *codes,
                          const int64_t *seq_off, const mauve_scoring *sc, int64_t band_from,
                          uint32_t *cols, int64_t *col_off, int64_t *score);

/* ---- extant sum-of-pairs score of ungapped matches: the anchor score behind ProgressiveAligner::setLcbScoringScheme(
        ExtantSumOfPairsScoring) (progressiveMauve.cpp:611-625; libMems-internal, frozen form DESIGN.md S11): for every column of a
        match and every pair of its components, the substitution score of the two bases.  length[n], start[n*nseq] signed
        1-based on the resident genomes, scores[n] out. ---- */
int mauve_match_sp_scores(mauve_ctx *ctx, int64_t n, const int64_t *length, const int64_t *start, const mauve_scoring *sc,
                          int64_t *scores);

/* ---- whole path: doAlignment's hot section (mauveAligner.cpp:523-531,585,629-698,746-760):
        multi-MUMs -> N-way filter -> overlap elimination -> LCBs -> recursive anchoring -> gapped
        alignment of every inter-anchor interval -> interval table.  Results are held by the context
        until the next call; fetch them with mauve_align_fetch.  Any fetch pointer may be NULL. ---- */
int mauve_align(mauve_ctx *ctx, const mauve_params *p, mauve_align_sizes *sizes);
/* The same with the match list the caller holds (Aligner::align(MatchList&, ...), mauveAligner.cpp:698: the list may come
   from a file, mauveAligner.cpp:484-503, or from a finder of the caller's own): no seed pass; the N-way matches of the
   list (all nseq starts non-zero; the others are ignored) are put in canonical order, overlap-eliminated and chained.
   length[n], start[n*nseq] signed 1-based. */
int mauve_align_matches(mauve_ctx *ctx, const mauve_params *p, int64_t n, const int64_t *length, const int64_t *start,
                        mauve_align_sizes *sizes);
/* The same resumed from LCBs the caller holds (an IntervalList read back from the .mln file of an earlier run, mauveAligner.cpp:
   705-722 --lcb-input; the matches of one LCB given to Aligner::align again, :723-744 --realign-lcb): anchor i has length[i],
   start[i*nseq + g] (signed 1-based, a component in every genome, forward in genome 0) and belongs to LCB lcb[i] (ids 0 .. n_lcb-1).
   The anchors of an LCB must form one collinear, overlap-free chain.  No seed pass, no overlap / breakpoint elimination, no LCB
   extension: recursive anchoring (p->recursive) and the gapped alignment of every inter-anchor interval (p->gapped) run as in
   mauve_align; n_mums of the result is 0. */
int mauve_align_lcbs(mauve_ctx *ctx, const mauve_params *p, int64_t n, const int64_t *length, const int64_t *start, const int64_t *lcb,
                     mauve_align_sizes *sizes);
int mauve_align_fetch(mauve_ctx *ctx,
                      int64_t *mum_length, int64_t *mum_start,            /* [n_mums], [n_mums*nseq] */
                      int64_t *lcb_left, int64_t *lcb_right, int64_t *lcb_weight, /* [n_lcb*nseq] x2, [n_lcb] */
                      int64_t *anchor_length, int64_t *anchor_start, int64_t *anchor_lcb,
                      int64_t *iv_left, int64_t *iv_right, int8_t *iv_reverse,   /* [n_iv*nseq] */
                      int64_t *col_off, uint32_t *cols, int64_t *dp_score);     /* [n_iv+1],[n_cols],[n_iv] */
/* The same result in the narrowest types that hold it (no reference counterpart: libMems hands out objects; this is what a caller that
   wants the arrays moves).  cols: col_bytes bytes per column (1: up to 8 genomes, 2: up to 16, 4: any; MAUVE_ERR_ARG when a column does not
   fit); match and anchor tables as int32 (a context holds fewer than 2^31 bases, so every start and length fits); the small per-LCB and
   per-interval tables as in mauve_align_fetch.  From page-locked buffers (mauve_host_alloc) the bulk arrays are narrowed on the device and
   copied once; any pointer may be NULL.  mauve_align_fetch is unchanged. */
int mauve_align_fetch_compact(mauve_ctx *ctx, int col_bytes,
                              int32_t *mum_length, int32_t *mum_start,
                              int64_t *lcb_left, int64_t *lcb_right, int64_t *lcb_weight,
                              int32_t *anchor_length, int32_t *anchor_start, int32_t *anchor_lcb,
                              int64_t *iv_left, int64_t *iv_right, int8_t *iv_reverse,
                              int64_t *col_off, void *cols, int64_t *dp_score);
/* ---- the same path in three phases, for sharding the gapped alignment of ONE alignment over several GPUs
        (mauveAligner.cpp:130-131 --realign-lcb "for parallelization of LCB alignment").  Every rank
        calls mauve_align_begin (deterministic: identical anchors and interval table everywhere), aligns its
        share of the n_dp intervals with mauve_align_dp (indices into the interval table; outputs compact, in
        idx order: cols needs sum of max_cols over idx, col_off n+1, score n), exchanges the columns (one
        all_gather, mauvealigner_amd/parallel.py) and calls mauve_align_finish with the columns of ALL
        intervals in table order.  mauve_align_dp_cost gives per-interval DP cells (for LPT packing) and the
        column capacity each interval needs. ------------------------------------------------------------- */
int mauve_align_begin(mauve_ctx *ctx, const mauve_params *p, int64_t *n_dp, int64_t *n_codes);
int mauve_align_begin_matches(mauve_ctx *ctx, const mauve_params *p, int64_t n, const int64_t *length, const int64_t *start,
                              int64_t *n_dp, int64_t *n_codes);            /* mauve_align_begin on the caller's match list */
int mauve_align_dp_cost(mauve_ctx *ctx, int64_t *cost, int64_t *max_cols);
/* the two anchors that flank every DP interval, records of (1 + nseq): length, signed starts -- what a GappedAligner
   other than the built-in one is called with (GappedAligner::Align(cr, left match, right match, seq_table),
   MatchRecord.h:311): left[n_dp*(1+nseq)], right[n_dp*(1+nseq)] */
int mauve_align_dp_anchors(mauve_ctx *ctx, int64_t *left, int64_t *right);
int mauve_align_dp(mauve_ctx *ctx, const int64_t *idx, int64_t n, uint32_t *cols, int64_t *col_off,
                   int64_t *score, int64_t *cells);
int mauve_align_finish(mauve_ctx *ctx, const uint32_t *cols, const int64_t *col_off, const int64_t *score,
                       int64_t cells, mauve_align_sizes *sizes);
/* ---- the same work spread over several contexts, one per GPU, without splitting the call: every rank holds the same genomes and
        makes the SAME calls (mauve_align, mauve_progressive_align, mauve_guide_tree); inside them the independent units SURVEY.md 8e
        names -- the pairwise finder passes of the guide tree (progressiveMauve.cpp:490-501), the gaps of every recursion level
        (mauveAligner.cpp:130-131: "for parallelization of LCB alignment"), the gapped-alignment intervals of a guide-tree node --
        are LPT-partitioned over the ranks, each rank works on its share, and the (small) results are exchanged through the
        caller's all-gather, so that every rank ends with the whole result, bit-identical to a single context's.  The seed pass
        over the whole genomes and the breakpoint elimination are not partitioned (every rank runs them).
        allgather(user, send, send_bytes, &recv, recv_bytes[world]): blocking; on return *recv points to the contributions of all
        ranks one behind the other in rank order (valid until the next call), recv_bytes[r] their sizes; returns 0 on success.
        world <= 1 or fn == NULL switches the sharding off. ---- */
typedef int (*mauve_allgather_fn)(void *user, const void *send, int64_t send_bytes, const void **recv, int64_t *recv_bytes);
int mauve_set_shard(mauve_ctx *ctx, int rank, int world, mauve_allgather_fn fn, void *user);
/* The same with RCCL inside the library: nccl_comm is the caller's ncclComm_t (one rank per GPU, created with ncclCommInitRank on the context's
   device).  Every exchange is one ncclAllGather of the ranks' sizes and one of the padded payloads, issued by the library on its own stream
   between device buffers; nothing goes through a callback.  RCCL is resolved at this call from what the process has loaded (the librccl the
   caller's communicator comes from; librccl.so.1 is opened if none is), so the library itself does not link it.  world == 1 switches the
   sharding off like mauve_set_shard (with MAUVE_SHARD_SINGLE set the one rank still runs every exchange: a single-GPU rehearsal of the path).
   mauve_shard_get_stats: exchanges, bytes sent and received, and milliseconds spent in them since the collective was set. */
int mauve_set_shard_rccl(mauve_ctx *ctx, int rank, int world, void *nccl_comm);
typedef struct { int64_t exchanges, bytes_sent, bytes_received; double ms; } mauve_shard_stats;
int mauve_shard_get_stats(mauve_ctx *ctx, mauve_shard_stats *out);

/* ---- progressiveMauve path: guide tree + ProgressiveAligner::align(seq_table, interval_list)
        (progressiveMauve.cpp:575-710; distance matrix / guide tree mauveAligner.cpp:616-623).  Frozen replacement
        ("guide-tree recursive anchoring"): UPGMA over pairwise-match coverage; the root aligns
        what all genomes share, every node below aligns -- among its own genomes -- the bases no ancestor has
        placed.  tree_left/right: [2*nseq-1] child ids (-1 for leaves, internal ids nseq.. in merge order);
        dist: [nseq*nseq] distances in parts per million; any of them may be NULL.  The result is fetched with
        mauve_align_fetch / mauve_write_xmfa: intervals with >= 2 genomes first (n_lcb of them), then the
        single-genome leftovers; absent genomes have left = right = 0. ------------------------------------- */
int mauve_guide_tree(mauve_ctx *ctx, uint64_t pattern, int64_t *dist, int32_t *tree_left, int32_t *tree_right);
/* The pairwise breakpoint estimate behind ProgressiveAligner::setBreakpointDistanceScale / setBpDistEstimateMinScore
   (progressiveMauve.cpp:628-642; the estimate is libMems-internal, frozen form DESIGN.md S11c): for every genome pair, the
   adjacencies between its pairwise matches of length >= min_len (ordered along the lower genome) that are not conserved
   in the other genome.  bp: [nseq*nseq], symmetric, zero diagonal. */
int mauve_breakpoint_counts(mauve_ctx *ctx, uint64_t pattern, int64_t min_len, int64_t *bp);
int mauve_progressive_align(mauve_ctx *ctx, const mauve_params *p, mauve_align_sizes *sizes,
                            int32_t *tree_left, int32_t *tree_right, int64_t *dist);
/* ProgressiveAligner::setInputGuideTreeFileName (progressiveMauve.cpp:689-690): the same alignment along the caller's
   tree instead of the UPGMA one.  tree_left/right: [2*nseq-1] in the form mauve_guide_tree returns -- leaves 0..nseq-1
   with -1, every internal node with two distinct children of smaller id, root last; anything else is MAUVE_ERR_ARG. */
int mauve_progressive_align_tree(mauve_ctx *ctx, const mauve_params *p, mauve_align_sizes *sizes,
                                 const int32_t *tree_left, const int32_t *tree_right);
/* ---- backbone and islands: detectBackbone(iv_list, bb_list, &BigGapsDetector(island_gap_size)) of applyBackbone
        (progressiveMauve.cpp:226-260, --island-gap-size :268) and simpleFindIslands / simpleFindBackbone
        (mauveAligner.cpp:822,844-845).  libMems-internal; frozen form DESIGN.md S12: per genome pair, a run of more
        than island_gap_size columns in which only one of the two has residues is an island; the pair is joined
        outside the gap regions that hold an island or touch an end of the interval; a backbone segment is a connected
        component (>= 2 genomes) of the joined pairs over a maximal run of columns.
        mauve_backbone works on the alignment the context holds (after mauve_align* / mauve_progressive_align*, columns
        still in HBM); mauve_backbone_alignment on the caller's (--apply-backbone, progressiveMauve.cpp:367-385).
        Fetch: seg_iv/seg_col/seg_len[n_seg] (interval, first column in it, columns), seg_mask[n_seg] genomes,
        seg_left/seg_right[n_seg*nseq] signed ends (negative = reverse strand, 0 = not in the segment);
        islands[n_islands*8]: interval, a, b (a < b), the genome that has the residues, first column, last column,
        its signed left and right end.  Order: interval, column, genome set / pair.  Any pointer may be NULL. ---- */
int mauve_backbone(mauve_ctx *ctx, int64_t island_gap_size, int64_t *n_seg, int64_t *n_islands);
int mauve_backbone_alignment(mauve_ctx *ctx, int nseq, int64_t n_iv, const int64_t *left, const int64_t *right,
                             const int8_t *reverse, const int64_t *col_off, const uint32_t *cols,
                             int64_t island_gap_size, int64_t *n_seg, int64_t *n_islands);
/* The homology pass in front of the backbone: detectAndApplyBackbone(iv_list, bb_list, hmm_params) (progressiveMauve.cpp:226-243; its
   HomologyHMM is libMems-internal, frozen form DESIGN.md S12b).  A two-state Viterbi path (homologous / unrelated) per interval and
   genome pair over the resident alignment's columns, in integer log-odds x 1000; residues that are homologous to no other genome of
   their column move to columns of their own.  The context's alignment is rewritten in place (column count and offsets change: *sizes
   receives the new sizes for mauve_align_fetch); n_moved = residues taken out of multi-genome columns.  Call before mauve_backbone. */
typedef struct { int32_t match, mismatch, gap, go_homologous, go_unrelated; } mauve_hmm_params;
/* the call site's knobs (progressiveMauve.cpp:319-322: identity 0.7, pgh 1e-5, pgu 1e-9) as integer scores: match = 1000 ln(id / .25),
   mismatch = 1000 ln((1 - id) / .75), gap = -500, transitions = 1000 ln(p).  identity outside (0.25, 1) or a probability outside (0, 1]
   has no such score: *h then carries positive transition scores, which mauve_apply_homology* refuse with MAUVE_ERR_ARG */
void mauve_hmm_params_from(double identity, double pgh, double pgu, mauve_hmm_params *h);
int mauve_apply_homology(mauve_ctx *ctx, const mauve_hmm_params *h, mauve_align_sizes *sizes, int64_t *n_moved);
/* ... of the caller's alignment (the genomes it refers to are the context's): cols_out holds up to one column per residue, col_off_out [n_iv+1].
   The column array must be consistent with the interval table -- right - left + 1 residues of every present genome, none of an absent one, no bit
   at or above nseq -- or the call returns MAUVE_ERR_ARG (mauve_backbone_alignment checks the same) */
int mauve_apply_homology_alignment(mauve_ctx *ctx, int nseq, int64_t n_iv, const int64_t *left, const int64_t *right, const int8_t *reverse,
                                   const int64_t *col_off, const uint32_t *cols, const mauve_hmm_params *h, int64_t *col_off_out, uint32_t *cols_out,
                                   int64_t *n_moved);
int mauve_backbone_fetch(mauve_ctx *ctx, int64_t *seg_iv, int64_t *seg_col, int64_t *seg_len, uint32_t *seg_mask,
                         int64_t *seg_left, int64_t *seg_right, int64_t *islands);
/* IntervalList::WriteStandardAlignment (mauveAligner.cpp:746-760; format mfa2xmfa.cpp:64-115).
   Two-phase: buf == NULL returns the needed size (including NUL) in *len. */
int mauve_write_xmfa(mauve_ctx *ctx, const char *const *names, char *buf, int64_t *len);

/* ---- measurement -------------------------------------------------------------------------------
   Per-kernel HIP-event timing on the context's stream (bench.py roofline).  Kernel ids: */
#define MAUVE_K_EXTRACT 0
#define MAUVE_K_SORT_HIST 1
#define MAUVE_K_SORT_SCAN 2
#define MAUVE_K_SORT_SCATTER 3
#define MAUVE_K_JOIN 4
#define MAUVE_K_EXTEND 5
#define MAUVE_K_DP 6
#define MAUVE_K_RUNS 7
#define MAUVE_K_CANON 8           /* canonical order on the device: key build, its (small) radix sort, gather */
#define MAUVE_K_MISC 9            /* the small sorts of the device chain and of the DP front end */
#define MAUVE_K_COUNT 10
int mauve_profile_enable(mauve_ctx *ctx, int on);
int mauve_profile_reset(mauve_ctx *ctx);
int mauve_profile_get(mauve_ctx *ctx, int kernel, double *total_ms, int64_t *launches, int64_t *units);
/* wall-clock of the stages of the last mauve_align / mauve_progressive_align call, milliseconds (progressive: summed over the
   nodes of the guide tree; tree_ms = the pairwise passes and the guide tree in front of them) */
typedef struct {
    double seed_ms, chain_ms, recurse_ms, dp_ms, assemble_ms, total_ms, tree_ms;
} mauve_stage_times;
int mauve_last_stage_times(mauve_ctx *ctx, mauve_stage_times *t);

#ifdef __cplusplus
}
#endif
#endif
