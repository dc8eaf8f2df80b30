// common.hpp -- internal declarations shared by the translation units of libmauve_hip.so.
// Product code: must never include anything from oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include <functional>
#include "workers.hpp"
#include <chrono>
#include <algorithm>

#include "../../include/mauve_hip.h"

#define HIPCHK(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__) + " (" __FILE__ ":" + \
                         std::to_string(__LINE__) + ")";                                          \
            return MAUVE_ERR_HIP;                                                                 \
        }                                                                                         \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = bytes + bytes / 8 + 4096;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    // grow keeping the first `used` bytes (the stream is drained first: work queued on it may still read the old block)
    hipError_t ensure_keep(size_t bytes, size_t used, hipStream_t stream)
    {
        if (bytes <= cap) return hipSuccess;
        void *np = nullptr;
        const size_t want = bytes + bytes / 2 + 4096;
        hipError_t e = hipMalloc(&np, want);
        if (e != hipSuccess) return e;
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) { (void)hipFree(np); return e; }
        if (p && used && (e = hipMemcpy(np, p, used, hipMemcpyDeviceToDevice)) != hipSuccess) { (void)hipFree(np); return e; }
        if (p) (void)hipFree(p);
        p = np; cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Page-locked host buffer: the destination of the larger device-to-host copies (a pageable destination goes
// through the runtime's staging buffers at a fraction of the link rate).  Grows, never shrinks; owned by the ctx.
struct PinnedBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipHostFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = bytes + bytes / 8 + 4096;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Seed pattern decomposed into runs of contiguous care positions (kernel argument, by value).
// K' ("digit-reversed forward mer"): care offset t_j contributes code << 2j, so that
//   reverse-complement mer = ~K' (masked) and forward mer = digit_reverse(K').
struct SeedShape {
    int span, weight, nruns;
    uint64_t keymask;          // (1 << 2*weight) - 1
    uint8_t run_src[32];       // bit offset of the run in the window (2 * first care offset)
    uint8_t run_bits[32];      // 2 * run length
    uint8_t run_dst[32];       // bit offset in K'
    uint64_t care_lo, care_hi; // 2-bit-expanded care mask of the window (offset t -> bits 2t, 2t+1)
};

// The genomes as the kernels see them: one packed buffer, windows numbered globally.
struct GenomeTab {
    int nseq;
    uint32_t gpos_off[MAUVE_MAX_SEQ + 1];   // first global window index of genome g
    uint32_t nwin[MAUVE_MAX_SEQ];           // valid window starts of genome g
    uint64_t word_off[MAUVE_MAX_SEQ];       // first 64-bit word of genome g in the packed buffer
    uint64_t mask_off[MAUVE_MAX_SEQ];       // first 64-bit word of genome g in the placed-base bitmap (S9), if any
};

// A set of packed genomes resident on the device (the main genomes, or the gap sub-sequences of one
// recursive-anchoring level).
struct GenomeSet {
    DevBuf *buf = nullptr;
    int nseq = 0;
    std::vector<int64_t> lens;
    std::vector<uint64_t> word_off;
    // guide-tree recursive anchoring: 1 bit per base, set = already placed by an ancestor node;
    // windows touching a set bit are invalid for seeding and extension.  nullptr = no mask.
    DevBuf *vmask = nullptr;
    // contig starts (mauve_set_genomes_contigs): 1 bit per base, set = a contig begins here; a window may not run
    // across such a base.  Same word layout as vmask (mask_off).  nullptr = single-contig sequences.
    DevBuf *cmask = nullptr;
    std::vector<uint64_t> mask_off;
};

// seed hits handed in by a host-side finder: n records of (1 + nseq) words {component set, value of every genome}
struct HostHits { uint32_t n; const uint32_t *rec; };

// SeedMatchEnumerator on the device: rule in, CSR result out (mult / start_off / starts may be null: counts only)
struct EnumRequest { int64_t min_multi, max_multi; int direct_only; int64_t n, ns; int64_t *mult, *start_off, *starts; };

struct AlignResult {
    mauve_align_sizes sz{};
    std::vector<int64_t> mum_length, mum_start;
    std::vector<int64_t> lcb_left, lcb_right, lcb_weight;
    std::vector<int64_t> anchor_length, anchor_start, anchor_lcb;
    std::vector<int64_t> iv_left, iv_right;
    std::vector<int8_t> iv_reverse;
    std::vector<int64_t> col_off;
    std::vector<uint32_t> cols;        // capacity buffer: the first n_cols entries are the result (see pipeline.cpp)
    size_t n_cols = 0;
    uint32_t cols_fill = 0;            // value every word outside cols_dirty holds (0 = no such invariant)
    std::vector<std::pair<size_t, size_t>> cols_dirty;   // (offset, length) ranges holding gap columns
    std::vector<int64_t> dp_score;
    // device-assembled result (assemble_dev.hip): the columns (res_cols), the anchor table and maybe the match list are still in HBM
    bool stale = false;                     // the genomes were replaced after this result was made: a fetch is refused (mauve_set_genomes)
    bool genomes_replaced = false;          // ... also set when the result was wholly on the host already: it can still be fetched, but the calls that read
                                            // bases against it (mauve_apply_homology, mauve_write_xmfa) are refused -- they would pair the new genomes with the old alignment
    bool dev_pending = false;               // anchor table / match list still on the device
    bool cols_pending = false;              // columns only in res_cols (cols_ext not set)
    size_t dev_na = 0, dev_nm = 0;          // anchors; matches still on the device (0: mum_* are filled)
    const int32_t *dev_alen = nullptr, *dev_ast = nullptr, *dev_alcb = nullptr;     // ... where the anchors are (chain_order_device's arrays)
    const uint32_t *cols_ext = nullptr;      // the columns in page-locked staging after materialize_result (else: cols)
    const uint32_t *cols_data() const { return cols_ext ? cols_ext : cols.data(); }
};

// N-way match list in flat records of (1 + N) int64: length, signed 1-based starts (libMems Match layout).
struct MatchVec {
    int N = 0;
    std::vector<int64_t> d;
    explicit MatchVec(int n = 0) : N(n) {}
    size_t size() const { return d.size() / (size_t)(1 + N); }
    bool empty() const { return d.empty(); }
    int64_t &len(size_t i) { return d[i * (1 + N)]; }
    int64_t len(size_t i) const { return d[i * (1 + N)]; }
    int64_t *st(size_t i) { return &d[i * (1 + N) + 1]; }
    const int64_t *st(size_t i) const { return &d[i * (1 + N) + 1]; }
    const int64_t *rec(size_t i) const { return &d[i * (1 + N)]; }
    void push(const int64_t *r) { d.insert(d.end(), r, r + 1 + N); }
    void push(int64_t l, const int64_t *starts) { d.push_back(l); d.insert(d.end(), starts, starts + N); }
    void resize(size_t n) { d.resize(n * (1 + N)); }
    void move(size_t dst, size_t src) { if (dst != src) std::copy(d.begin() + src * (1 + N), d.begin() + (src + 1) * (1 + N), d.begin() + dst * (1 + N)); }
    void reserve(size_t n) { d.reserve(n * (1 + N)); }
    void sort_by_start0();
};
struct DpSeqDesc { int32_t genome; int32_t rev; int64_t lo0; int64_t len; };   // lo0: 0-based left end in the genome

// state of an alignment between its begin and finish phases (pipeline.cpp)
struct AlignState {
    struct GapRef { int64_t lcb, idx; bool dp; int64_t dp_slot; int64_t tot; };
    struct Item { int64_t lcb; uint32_t idx; int64_t col0; int64_t gap; };
    bool open = false;
    bool anchor_table_done = false;     // anchor_length/start/lcb already filled (in the DP kernel's shadow)
    bool dev_tail = false;              // the chains stayed on the device: DP front end and assembly run there (mauve_align)
    const int32_t *dv_len = nullptr, *dv_st = nullptr, *dv_lcb = nullptr;   // ... where: anchors in chain order (chain_order_device, or the extended list)
    bool lw_from_host = false;          // the LCB weights of the result are those align_begin left in R.lcb_weight (device extension)
    int64_t mums_kept = -1;             // >= 0: the main pass's match list (that many records) sits in ctx->sorted_rec_keep while the recursion's passes use sorted_rec
    mauve_params p{};
    int N = 0; uint32_t full = 0;
    int64_t sum = 0, nm = 0, nl = 0, n_dp = 0, code_total = 0, n_anchor = 0, anchor_cols = 0;
    double t0 = 0, t_dp0 = 0;
    std::vector<MatchVec> chains;
    std::vector<GapRef> gaps;
    std::vector<DpSeqDesc> desc;
    std::vector<int64_t> dcol_off, dscore;       // DP columns of mauve_align land in ctx->pin_dcols
    // scratch of the host stages; like everything above it keeps its capacity from call to call -- a fresh
    // megabyte-sized vector per stage and call costs more in page faults than the stage itself
    MatchVec m;
    std::vector<int64_t> match_lcb;
    std::vector<int64_t> match_weight;  // sum-of-pairs scores of the matches (lcb_scoring = SP), else empty
    std::vector<Item> items;
    // start a new alignment: scalars to zero, vectors emptied but not released
    void reset()
    {
        open = false; anchor_table_done = false; dev_tail = false; dv_len = dv_st = dv_lcb = nullptr; lw_from_host = false; mums_kept = -1; p = mauve_params(); N = 0; full = 0;
        sum = nm = nl = n_dp = code_total = n_anchor = anchor_cols = 0; t0 = t_dp0 = 0;
        gaps.clear(); desc.clear(); dcol_off.clear(); dscore.clear();
        match_lcb.clear(); match_weight.clear(); items.clear();
    }
};

struct mauve_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;           // the one-wave DP launch runs here when there are workgroup launches (those go first, on `stream`)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::string err;
    char devname[256] = {0};
    int cus = 256;                       // compute units of the device (launches sized to a whole number of waves per SIMD)

    // genomes
    int nseq = 0;
    std::vector<int64_t> lens;
    std::vector<uint64_t> word_off;      // per genome, in 64-bit words
    std::vector<const uint64_t *> host_packed;   // host copy of every genome's packed words (XMFA text), inside pin_genomes
    PinnedBuf pin_genomes;
    PinnedBuf pin_tail;                  // direct upload from page-locked caller memory: the fixed-up tail words of every genome
    bool host_copy_valid = false;         // host_packed / pin_genomes hold the genomes (else: fetched back on demand, host_genomes)
    size_t total_words = 0;
    DevBuf genomes;
    // ambiguous bases and contig starts of the resident genomes (mauve_set_genomes_contigs): device bitmaps in the
    // layout of GenomeSet::vmask / cmask, their host copies (masks of the guide-tree nodes and of the LCB extension
    // are built on top of the first; the XMFA writer prints N where it is set)
    DevBuf base_invalid, contig_mask, node_cmask;
    bool has_invalid = false, has_contigs = false;
    std::vector<uint64_t> h_invalid, h_contig, base_mask_off;

    // seed-pass workspace
    DevBuf keysA, keysB, valsA, valsB, hist, totals, posmask, hit_mask, hit_pos, hit_seg, cand, mlen, mstart, counters;
    DevBuf sorted_rec;                   // canonical order on the device: gathered (length, starts) records as int64
    DevBuf canon_k1, canon_k2, canon_v1, canon_v2;   // ... and its (key, candidate index) sort buffers
    DevBuf join_bound;                   // join_hash: first bucket boundary at or after every chunk edge
    DevBuf join_ovf;                     // join_hash: [count, pad, (lo, hi) ...] ranges handed back to the full sort + serial join
    DevBuf ch_len, ch_st, ch_crop, ch_ent, ch_ord, ch_rank, ch_node, ch_graph, ch_cnt;   // device chain (chain_dev.hip)
    DevBuf gap_work;                     // recursion batches: collinear rule and survivor compaction on the device (chain_device_gaps_compact)
    DevBuf ch_big;                       // working arrays of overlap clusters beyond the per-thread limit (recursion batches)
    DevBuf ch_anch, ch_lw;               // the chains in chain order and the LCB weights (chain_order_device)
    DevBuf ch_anch2, ext_work, sorted_rec_keep;   // device-resident LCB extension (extend_dev.hip): the extended anchor list, its work area, the main pass's match list set aside
    PinnedBuf pin_ext;
    DevBuf ch_mw;                        // sum-of-pairs scores of the chain's cropped records (score-weighted LCBs on the device chain)
    DevBuf as_wide;                      // the anchor table widened to int64 for a direct fetch (compact fetch: the match list narrowed to int32)
    DevBuf res_narrow;                   // the columns narrowed to 8 / 16 bits for a compact fetch
    DevBuf hom_cols;                     // homology pass (backbone_dev.hip): the re-split columns, swapped with res_cols when done
    DevBuf as_work, as_isl, res_cols;    // device assembly (assemble_dev.hip): work area, islands, result columns
    PinnedBuf pin_tab;                   // anchor table and match list of a device-assembled result on their way to the host
    PinnedBuf pin_asm, pin_cols;         // ... its per-LCB rows coming back; the columns and anchors on their way to a fetch
    PinnedBuf pin_chain;
    PinnedBuf pin_mask;                  // LCB extension: the valid-piece bitmap on its way to placed_mask
    // backbone / islands of the last alignment (backbone_dev.hip)
    struct BackboneResult {
        int N = 0; bool valid = false;
        std::vector<int64_t> seg_iv, seg_col, seg_len, seg_left, seg_right, islands;
        std::vector<uint32_t> seg_mask;
    } bb;
    PinnedBuf pin_bb;                    // rank queries and their answers
    DevBuf bb_cols, bb_work, bb_query;   // a caller's / a host-assembled column array; interval table + records; rank queries
    size_t bb_rec_cap = 0;
    DevBuf run_sum;                      // pairwise finder: run list (start, length, exactly-once genome set)
    DevBuf rec_vinv, rec_vcm;            // ... and their ambiguity / contig bitmaps, when the resident genomes have them
    DevBuf rec_genomes, rec_seg;         // recursive anchoring: gap sub-sequences + segment table
    DevBuf placed_mask;                  // guide-tree recursive anchoring: placed-base bitmap
    // last match list (canonical order, host) + nseq it refers to
    std::vector<int64_t> match_len, match_start;
    int64_t n_matches = 0;
    EnumRequest *enum_req = nullptr;       // set for the duration of mauve_seed_match_enumerate
    const HostHits *host_hits = nullptr;  // set for the duration of mauve_extend_hits
    int64_t dev_rec_n = -1;              // >= 0: sorted_rec holds that many records (int64 length[n], start[n*nseq]) in canonical order

    // DP workspace
    DevBuf dp_desc, dp_list, dp_codes, dp_off, dp_prof_cnt, dp_prof_mask, dp_prof2_cnt, dp_prof2_mask, dp_tb, dp_meta, dp_score,
        dp_cols, dp_rows, dp_sp, dp_pick, dp_wflags;

    // the seed pass may leave its match list on the device only (sorted_rec) when the caller says so: mauve_align's device tail
    bool pair_sums_only = false;          // seed pass for the guide tree: per-pair length sums instead of the match list
    std::vector<int64_t> pair_sums;
    int64_t bp_min_len = -1;              // >= 0 with pair_sums_only: also count the broken adjacencies of every pair's matches of at least this length
    std::vector<int64_t> pair_bp;         // [N*N], upper triangle (a < b)
    DevBuf bp_work;
    std::vector<uint8_t> rec_flags;       // one byte per anchor in chain order (LCB by LCB): the gap behind it is a candidate of the recursion; empty = not known (recursive.cpp tests every gap)
    bool lazy_matches_ok = false, matches_pending = false;
    int match_nseq = 0;
    // where dp_run_from_anchors left its device-side results (valid until the next DP launch)
    struct DpFrontOut { const int32_t *alen, *ast, *alcb, *gapcode; const int64_t *col_off, *score; const uint32_t *cols; int64_t n_dp, n_cols; } dpf_out{};
    std::vector<DpSeqDesc> prog_desc;      // progressive.cpp: the descriptors of a node's intervals and of their refinement candidates (kept: no fresh pages per node)
    const int64_t *dp_last_col_off = nullptr; int64_t dp_last_n = 0;     // column offsets (device) and size of the batch dp_core ran last: dp_fetch_picked
    int64_t dp_band_from = INT64_MAX;     // intervals whose longest sequence exceeds this run the banded DP (dp_batch.hip)
    DevBuf dpf_anch, dpf_work, dpf_tot;   // device front end of the DP stage (dp_run_from_anchors)

    // profiling
    bool prof = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double k_ms[MAUVE_K_COUNT] = {0};
    int64_t k_launch[MAUVE_K_COUNT] = {0};
    int64_t k_units[MAUVE_K_COUNT] = {0};

    PinnedBuf pin_anch;                  // anchors in chain order on their way to the device, and the gap codes coming back
    PinnedBuf pin_dcols;                 // DP columns of the whole-call path (mauve_align, mauve_progressive_align)
    PinnedBuf pin_meta;                  // per-interval DP results (length, score, cells)
    PinnedBuf pin_dp_in;                 // DP inputs on their way to the device: offset tables, interval list, descriptors
    PinnedBuf pin_seed;                  // seed pass: counter readback (first 64 B) and the candidates' match records
    // host scratch of dp_core, kept across calls (see AlignState)
    struct DpHost {
        std::vector<int64_t> tb_off, rows_off, est, need, nmax, lst, lst2, seq_off, tb_list;
        std::vector<uint8_t> is_big, cls;
    } dph;
    // host scratch of the seed pass (match records before the canonical sort)
    struct SeedHost {
        std::vector<int32_t> hl, hs;
        std::vector<uint32_t> order, tmp;
        std::vector<uint64_t> k1;
    } sdh;

    // host work that does not depend on the seed pass, run by the seed pass right before its first wait on the
    // stream (extract, sort, join and run detection are in flight by then); one shot
    std::function<void()> shadow;

    SpinPool *pool = nullptr;            // host helpers, armed for the duration of an align call (workers.hpp)

    // one alignment spread over several contexts (mauve_set_shard): this rank, their number, the caller's all-gather
    int shard_rank = 0, shard_world = 1; mauve_allgather_fn shard_fn = nullptr; void *shard_user = nullptr;
    bool shard_on = false;                // the independent units inside the calls are dealt out and exchanged (world > 1; a one-rank RCCL communicator with MAUVE_SHARD_SINGLE: the same code, one part)
    void *shard_comm = nullptr;           // ncclComm_t of mauve_set_shard_rccl: the library runs the all-gathers itself, on its stream, through device buffers
    DevBuf shard_dev;                     // device side of an RCCL exchange: [my size | all sizes | my payload | all payloads]
    PinnedBuf shard_pin;                  // ... and its page-locked host side
    mauve_shard_stats shard_stat = {0, 0, 0, 0.0};

    AlignResult res;
    AlignState ast;
    mauve_stage_times stage{};
};

// RAII-less helper: time one kernel launch on ctx->stream when profiling is on.
struct KernelTimer {
    mauve_ctx *c; int id; int64_t units;
    KernelTimer(mauve_ctx *ctx, int kid, int64_t u) : c(ctx), id(kid), units(u)
    {
        if (c->prof) (void)hipEventRecord(c->ev0, c->stream);
    }
    ~KernelTimer()
    {
        if (!c->prof) return;
        (void)hipEventRecord(c->ev1, c->stream);
        (void)hipEventSynchronize(c->ev1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
        c->k_ms[id] += ms; c->k_launch[id] += 1; c->k_units[id] += units;
    }
};

static inline double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// ---- internal entry points between translation units ----
// the contributions of all ranks to one exchange: parts[r] = (pointer, bytes), valid until the next exchange (api.cpp)
int rec_gap_flags_device(mauve_ctx *c, const int32_t *alen, const int32_t *ast, const int32_t *alcb, int64_t na, int N, int64_t min_gap, uint8_t *d_flag);
int shard_allgather(mauve_ctx *c, const void *send, size_t bytes, std::vector<std::pair<const char *, size_t>> &parts);
// deterministic LPT packing of `cost` into the ranks (ties: lower index first, lower rank first): owner[i] = rank of unit i
void shard_lpt(const std::vector<int64_t> &cost, int world, std::vector<int> &owner);
bool make_seed_shape(uint64_t pattern, SeedShape *out);
GenomeSet main_genome_set(mauve_ctx *ctx);
int seedpass_run(mauve_ctx *ctx, const GenomeSet &gs, uint64_t pattern, int mode, uint64_t mask, int extend,
                 const uint32_t *seg_dev, uint32_t nseg, int64_t *n_matches);
int seedpass_enumerate(mauve_ctx *ctx, const GenomeSet &gs, int seq, uint64_t pattern, EnumRequest &q);
int seedpass_from_hits(mauve_ctx *ctx, const GenomeSet &gs, uint64_t pattern, const HostHits &hits, int extend, int64_t *n_matches);
int seedpass_sorted_list(mauve_ctx *ctx, const GenomeSet &gs, int seq, uint64_t pattern, std::vector<uint64_t> *keys,
                         std::vector<uint32_t> *vals, int *weight);

int sort_pairs_u32(mauve_ctx *ctx, uint32_t n, int key_bits, uint32_t **keys_io, uint32_t **vals_io, uint32_t *keys_alt, uint32_t *vals_alt,
                   int timer_id);
// device chain (chain_dev.hip): EliminateOverlaps + LCBs of the N-way list the seed pass left in ctx->sorted_rec
int chain_device(mauve_ctx *c, int N, int64_t min_weight, bool collinear, MatchVec &m, std::vector<int64_t> &match_lcb, int64_t &n_lcb, const mauve_scoring *sp_scoring = nullptr);
struct ChainGraphHost { uint32_t na, K; int64_t *weight; uint32_t *orient; int32_t *prev, *next; int32_t *final_stage; int32_t *final_dev; };
int chain_device_graph(mauve_ctx *c, int N, int64_t maxlen, const uint32_t *seg0, uint32_t nseg, ChainGraphHost *G, bool graph_to_host = true, const mauve_scoring *sp_scoring = nullptr);
int chain_device_gaps_compact(mauve_ctx *c, int N, int64_t maxlen, const uint32_t *seg0_dev, uint32_t nseg, const int32_t **hl_out, const int32_t **hs_out,
                              const uint32_t **hgap_out, uint32_t *ns_out);
int chain_device_core(mauve_ctx *c, int N, int64_t min_weight, bool collinear, int64_t &n_lcb, const mauve_scoring *sp_scoring = nullptr);
int chain_device_gaps(mauve_ctx *c, int N, int64_t maxlen, const uint32_t *seg0_dev, uint32_t nseg, const int32_t **hl_out, const int32_t **hs_out,
                      std::vector<uint8_t> &survive);
int chain_device_copy_back(mauve_ctx *c, int N, MatchVec &m, std::vector<int64_t> &match_lcb);
int chain_order_device(mauve_ctx *c, int N, int64_t nl, int64_t min_gap, int64_t *na_out, int64_t *n_rec_out);
int extend_lcbs_device(mauve_ctx *c, const mauve_params *p, int w, int64_t lcbw, int N, const int32_t **alen_io, const int32_t **ast_io,
                       const int32_t **alcb_io, int64_t *na_io, int64_t *nl_io, int64_t *n_rec_io, std::vector<int64_t> &lcb_weight);

// host chaining (chain_host.cpp)
struct ChainOrders { std::vector<std::vector<uint32_t>> ord; bool sparse = false; };   // per genome: match indices in left-end order;
                                                                                        // sparse: the list still holds dead records (not named here)
void host_eliminate_overlaps(MatchVec &m, ChainOrders *orders = nullptr, bool compact = true);
int seed_family_matches(mauve_ctx *c, const GenomeSet &gs, int w, int mode, uint64_t mask, MatchVec &out);    // pipeline.cpp
void host_merge_matches(MatchVec &kept, const MatchVec &add);        // seed families: kept + what of add no kept match contains
void host_left_orders(const MatchVec &m, ChainOrders &orders);       // per-genome left-end order of an overlap-free list
void host_lcb_chain(const MatchVec &m, int64_t min_weight, bool collinear, std::vector<int64_t> &match_lcb, int64_t &n_lcb,
                    const ChainOrders *orders = nullptr, const int64_t *match_weight = nullptr);
// extant sum-of-pairs scores of the matches of m (n components; gmap: their genomes, nullptr = 0..n-1), assemble_dev.hip
int match_sp_scores(mauve_ctx *c, const MatchVec &m, const int *gmap, const mauve_scoring *sc, std::vector<int64_t> &out);
int64_t sp_default_min_weight(int w, int n, const mauve_scoring *sc);

void lcb_greedy(int N, int32_t K, int64_t *weight, const uint32_t *orient_bits, int32_t *prevv, int32_t *nextv, int64_t min_weight,
                bool collinear, std::vector<int64_t> &final_id, int64_t &n_lcb);

// DP (dp_batch.hip)
int dp_batch_run_desc(mauve_ctx *ctx, int nseq, int64_t n_iv, const DpSeqDesc *desc, const mauve_scoring *sc,
                      uint32_t *cols, int64_t *col_off, int64_t *score, int64_t *cells, bool may_shard = false, int64_t *sp = nullptr);
// the same with the rows behind n_orig made on the device as rotations of the first n_orig (cbase: first candidate of every interval, n_orig + 1 entries);
// only when dp_desc_rotations_on_device(n_total)
bool dp_desc_rotations_on_device(int64_t n_total);
int dp_batch_run_desc_rot(mauve_ctx *ctx, int nseq, int64_t n_orig, const DpSeqDesc *desc, const int32_t *cbase, int64_t n_total, const mauve_scoring *sc,
                          int64_t *col_off, int64_t *score, int64_t *cells, int64_t *sp);
// gapped-alignment eligibility of an inter-anchor interval by its longest sequence: full DP up to max_gapped_len, banded
// DP above it up to max_banded_len
inline int64_t dp_len_limit(const mauve_params *p) { return p->max_banded_len > p->max_gapped_len ? p->max_banded_len : p->max_gapped_len; }
inline int64_t dp_band_from_of(const mauve_params *p) { return p->max_banded_len > p->max_gapped_len ? p->max_gapped_len : INT64_MAX; }
int dp_run_from_anchors(mauve_ctx *ctx, int N, int64_t na, const int32_t *h_len, const int32_t *h_st, const int32_t *h_lcb, int gapped,
                        int64_t max_gapped_len, const mauve_scoring *scoring, int32_t *gapcode, int64_t *n_dp_out, int64_t *code_total_out,
                        PinnedBuf *dcols, std::vector<int64_t> &dcol_off, std::vector<int64_t> &dscore, int64_t *cells, int stay = 0);
int assemble_device(mauve_ctx *c, int64_t na, int64_t cells, mauve_align_sizes *sizes, bool host_chains = false);
int materialize_result(mauve_ctx *c);
int materialize_tables(mauve_ctx *c);
int fetch_columns(mauve_ctx *c, uint32_t *dst);
bool fetch_compact_direct(mauve_ctx *c, int col_bytes, int32_t *mum_length, int32_t *mum_start, int32_t *anchor_length, int32_t *anchor_start, int32_t *anchor_lcb,
                          void *cols, bool *tables_done, bool *cols_done, int *rc_out);
bool fetch_tables_direct(mauve_ctx *c, int64_t *mum_length, int64_t *mum_start, int64_t *anchor_length, int64_t *anchor_start, int64_t *anchor_lcb, int *rc_out);
bool host_pointer_is_pinned(const void *p);
int host_genomes(mauve_ctx *c);
int seed_matches_to_host(mauve_ctx *ctx);
int dp_fetch_picked(mauve_ctx *ctx, int64_t n_pick, const int64_t *pick, const int64_t *col_off, uint32_t *out, int64_t *out_off);
int dp_batch_run(mauve_ctx *ctx, int nseq, int64_t n_iv, const uint8_t *codes, const int64_t *seq_off,
                 const mauve_scoring *sc, uint32_t *cols, int64_t *col_off, int64_t *score, int64_t *cells);
