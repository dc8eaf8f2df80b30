// seed_pass.hip -- the seed pass of the hot path on gfx950:
//   seed_extract  : 2-bit packed genomes -> (canonical masked mer, global position | strand)
//   rs_*          : LSD radix sort of the pairs (the sorted mer list of all genomes at once)
//   mum_join      : runs of identical mers -> seed hits by the MatchFinder subclass rule
//   mum_candidates / mum_extend : ungapped extension of every hit into its maximal match
//
// Stands behind mems::MatchFinder::FindMatches [EXT] (call sites mauveAligner.cpp:523-589,
// progressiveMauve.cpp:490-501), with the in-tree rules of UniqueMatchFinder.cpp:36-60 and
// SeedMatchEnumerator.h:71-141.  Semantics are frozen in DESIGN.md S3/S4 and checked bit-exactly
// against oracle/ by tests/ (the oracle is never linked here).
//
// All kernels are HBM-bound integer work: coalesced 4/8-byte streams, LDS staging for
// the scatter, wave64 ballots for ranking and for the extension walk.  No MFMA by design.
#include "common.hpp"
#include "dev_scan.hpp"
#include <algorithm>
#include <cstring>
#include <cstdlib>

static const bool g_trace = getenv("MAUVE_TRACE") != nullptr;
#define TRACE(ctx, label) do { if (g_trace) { (void)hipStreamSynchronize((ctx)->stream); double t__ = now_ms(); fprintf(stderr, "[trace] %-22s +%.3f ms\n", label, t__ - trace_t0); trace_t0 = t__; } } while (0)

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t bits_from128(uint64_t lo, uint64_t hi, int start, int nbits)
{
    uint64_t v;
    if (start >= 64) v = hi >> (start - 64);
    else v = start ? ((lo >> start) | (hi << (64 - start))) : lo;
    return nbits >= 64 ? v : (v & ((1ULL << nbits) - 1ULL));
}

// K' of the window starting at base p of the genome whose packed words start at G
__device__ __forceinline__ uint64_t kprime_at(const uint64_t *__restrict__ G, uint32_t p, const SeedShape &sh)
{
    uint32_t q = p >> 5; int r = (p & 31) * 2;
    uint64_t w0 = G[q], w1 = G[q + 1], w2 = G[q + 2];
    uint64_t lo = r ? ((w0 >> r) | (w1 << (64 - r))) : w0;
    uint64_t hi = r ? ((w1 >> r) | (w2 << (64 - r))) : w1;
    uint64_t k = 0;
    for (int i = 0; i < sh.nruns; i++)
        k |= bits_from128(lo, hi, sh.run_src[i], sh.run_bits[i]) << sh.run_dst[i];
    return k;
}

// placed-base bitmap: is any of the `span` bases from p on already placed?
__device__ __forceinline__ bool window_masked(const uint64_t *__restrict__ M, uint32_t p, int span)
{
    const uint32_t q = p >> 6; const int r = p & 63;
    const uint64_t m0 = M[q], m1 = M[q + 1];
    const uint64_t v = r ? ((m0 >> r) | (m1 << (64 - r))) : m0;
    return (v & ((span >= 64) ? ~0ULL : ((1ULL << span) - 1ULL))) != 0;
}

// Is the window starting at base p unusable?  VM: 1 bit per base, set = no window may touch it (an ancestor placed it,
// DESIGN.md S9; or it is an ambiguous base, mauve_set_genomes_contigs); CM: 1 bit per base, set = a contig starts
// here -- a window may begin at such a base but not run across it (RepeatHashCat.h:19-20).  Either may be null.
__device__ __forceinline__ bool window_blocked(const uint64_t *__restrict__ VM, const uint64_t *__restrict__ CM, uint32_t p, int span)
{
    bool b = false;
    if (VM) b = window_masked(VM, p, span);
    if (CM && span > 1) b |= window_masked(CM, p + 1, span - 1);
    return b;
}

// narrow form (span <= 32, weight <= 16): the window is one 64-bit word and K' fits 32 bits
__device__ __forceinline__ uint32_t kprime_narrow(const uint64_t *__restrict__ G, uint32_t p, const SeedShape &sh)
{
    const uint32_t q = p >> 5; const int r = (p & 31) * 2;
    const uint64_t w0 = G[q], w1 = G[q + 1];
    const uint64_t lo = r ? ((w0 >> r) | (w1 << (64 - r))) : w0;
    uint32_t k = 0;
    for (int i = 0; i < sh.nruns; i++)
        k |= ((uint32_t)(lo >> sh.run_src[i]) & ((1u << sh.run_bits[i]) - 1u)) << sh.run_dst[i];
    return k;
}

__device__ __forceinline__ uint32_t digit_reverse32(uint32_t k, int weight)
{
    uint32_t x = __brev(k) >> (32 - 2 * weight);
    return ((x & 0xAAAAAAAAu) >> 1) | ((x & 0x55555555u) << 1);
}

// the 2-bit window of `span` bases starting at base p, as a 128-bit little-endian digit string
__device__ __forceinline__ void window_at(const uint64_t *__restrict__ G, uint32_t p, uint64_t &lo, uint64_t &hi)
{
    uint32_t q = p >> 5; int r = (p & 31) * 2;
    uint64_t w0 = G[q], w1 = G[q + 1], w2 = G[q + 2];
    lo = r ? ((w0 >> r) | (w1 << (64 - r))) : w0;
    hi = r ? ((w1 >> r) | (w2 << (64 - r))) : w1;
}

__device__ __forceinline__ uint64_t digit_reverse64(uint64_t x)
{
    x = __brevll(x);
    return ((x & 0xAAAAAAAAAAAAAAAAULL) >> 1) | ((x & 0x5555555555555555ULL) << 1);
}

// reverse complement of a span-base window (digits reversed, complemented); bits above 2*span are garbage
// that the care mask removes
__device__ __forceinline__ void window_revcomp(uint64_t lo, uint64_t hi, int span, uint64_t &rlo, uint64_t &rhi)
{
    const uint64_t nlo = digit_reverse64(~hi), nhi = digit_reverse64(~lo);   // 128-bit digit reversal
    const int s = 128 - 2 * span;                                            // 30 <= s <= 126 (span 1..49)
    if (s >= 64) { rlo = nhi >> (s - 64); rhi = 0; }
    else { rlo = (nlo >> s) | (nhi << (64 - s)); rhi = nhi >> s; }
}

// reverse the order of the 2-bit digits of a 2*weight-bit value
__device__ __forceinline__ uint64_t digit_reverse(uint64_t k, int weight)
{
    uint64_t x = __brevll(k) >> (64 - 2 * weight);
    return ((x & 0xAAAAAAAAAAAAAAAAULL) >> 1) | ((x & 0x5555555555555555ULL) << 1);
}

// genome of a global window index.  gpos_off[i] for i > nseq is all ones (build_tab), so up to eight genomes need no
// loop: seven compares against values the scalar unit loads once per kernel.
__device__ __forceinline__ int genome_of(uint32_t gpos, const GenomeTab &t)
{
    int g = 0;
    if (t.nseq <= 8) {
#pragma unroll
        for (int i = 1; i < 8; i++) g += (gpos >= t.gpos_off[i]) ? 1 : 0;
    } else
        for (int i = 1; i < t.nseq; i++) g += (gpos >= t.gpos_off[i]) ? 1 : 0;
    return g;
}

// ------------------------------------------------------------------------------------------------
// seed_extract: one thread per window.  Reads 0.25 B/position (L1-shared), writes key + val.
// val = global window index | strand << 31.
// ------------------------------------------------------------------------------------------------
// Segmented mode (recursive anchoring): genome g is a concatenation of nseg gap
// sub-sequences; seg[g*(nseg+1)+k] is the first base of segment k.  A window is valid only inside one
// segment and its key is prefixed with the segment id, so mers only meet inside their own gap; invalid
// windows get the all-ones key, which the join ignores.
__device__ __forceinline__ uint32_t seg_of(const uint32_t *__restrict__ segs, uint32_t nseg, uint32_t p)
{
    uint32_t lo = 0, hi = nseg;            // segs[lo] <= p < segs[hi]
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (segs[mid] <= p) lo = mid; else hi = mid; }
    return lo;
}

template <typename KeyT, bool SEG>
__global__ void __launch_bounds__(256) seed_extract(const uint64_t *__restrict__ packed, GenomeTab tab,
                                                    SeedShape sh, int g, KeyT *__restrict__ keys,
                                                    uint32_t *__restrict__ vals, uint32_t out_base,
                                                    const uint32_t *__restrict__ seg, uint32_t nseg)
{
    uint32_t n = tab.nwin[g];
    const uint64_t *G = packed + tab.word_off[g];
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        uint64_t kp = kprime_at(G, p, sh);
        uint64_t f = digit_reverse(kp, sh.weight);
        uint64_t r = (~kp) & sh.keymask;
        uint32_t s = r < f;
        uint64_t key = s ? r : f;
        if (SEG) {
            const uint32_t *sg = seg + (size_t)g * (nseg + 1);
            uint32_t k = seg_of(sg, nseg, p);
            key = (p + sh.span <= sg[k + 1]) ? (((uint64_t)k << (2 * sh.weight)) | key) : ~0ULL;
        }
        keys[out_base + p] = (KeyT)key;
        vals[out_base + p] = (tab.gpos_off[g] + p) | (s << 31);
    }
}

// All genomes in one launch, tiled exactly like the radix sort (4096 windows per workgroup), with the digit
// histogram of the first sort pass accumulated on the way out: saves two launches and one re-read of the keys.
// K' of a window given as a 128-bit digit string (the run loop of kprime_at / kprime_narrow)
__device__ __forceinline__ uint64_t kprime_of(uint64_t lo, uint64_t hi, const SeedShape &sh)
{
    uint64_t k = 0;
    for (int i = 0; i < sh.nruns; i++) k |= bits_from128(lo, hi, sh.run_src[i], sh.run_bits[i]) << sh.run_dst[i];
    return k;
}
__device__ __forceinline__ uint32_t kprime_of32(uint64_t lo, const SeedShape &sh)
{
    uint32_t k = 0;
    for (int i = 0; i < sh.nruns; i++) k |= ((uint32_t)(lo >> sh.run_src[i]) & ((1u << sh.run_bits[i]) - 1u)) << sh.run_dst[i];
    return k;
}

template <typename KeyT, bool SEG, bool NARROW>
__global__ void __launch_bounds__(256) seed_extract_all(const uint64_t *__restrict__ packed, GenomeTab tab, SeedShape sh,
                                                        KeyT *__restrict__ keys, uint32_t *__restrict__ vals, uint32_t P,
                                                        const uint32_t *__restrict__ seg, uint32_t nseg,
                                                        uint32_t *__restrict__ hist, uint32_t nblk,
                                                        const uint64_t *__restrict__ vmask, int hist_shift,
                                                        const uint64_t *__restrict__ cmask)
{
    // A thread takes FOUR CONSECUTIVE windows at a time (four such groups: 4096 windows per workgroup, the sort's tile): their bases come from the
    // same three or four packed words (one set of loads instead of four), and the keys and values leave as 16-byte stores.  With one window per
    // thread and 4-byte stores the kernel ran at 1.6 TB/s, waiting on its own store issue (wait share 0.71), not on the arithmetic.
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * 4096u;
#pragma unroll 2
    for (int i = 0; i < 4; i++) {
        const uint32_t gp0 = base + i * 1024 + threadIdx.x * 4;
        if (gp0 >= P) break;
        const int g0 = genome_of(gp0, tab);
        const bool together = gp0 + 3 < P && gp0 + 3 < tab.gpos_off[g0 + 1];       // all four in the buffer and in one genome
        uint64_t key[4]; uint32_t sf[4];
        if (together) {
            const uint32_t p0 = gp0 - tab.gpos_off[g0];
            const uint64_t *G = packed + tab.word_off[g0];
            const uint32_t q = p0 >> 5; const int r0 = (p0 & 31) * 2;                 // p0 is a multiple of 4 only relative to gp0: r0 is any even offset
            const uint64_t w0 = G[q], w1 = G[q + 1], w2 = G[q + 2], w3 = NARROW ? 0ULL : G[q + 3];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int r = r0 + 2 * j;                                             // 0 .. 68
                uint64_t a0 = w0, a1 = w1, a2 = w2; int rr = r;
                if (r >= 64) { a0 = w1; a1 = w2; a2 = w3; rr = r - 64; }
                const uint64_t lo = rr ? ((a0 >> rr) | (a1 << (64 - rr))) : a0;
                if (NARROW) {
                    const uint32_t kp = kprime_of32(lo, sh);
                    const uint32_t f = digit_reverse32(kp, sh.weight), rv = (~kp) & (uint32_t)sh.keymask;
                    sf[j] = rv < f; key[j] = sf[j] ? rv : f;
                } else {
                    const uint64_t hi = rr ? ((a1 >> rr) | (a2 << (64 - rr))) : a1;
                    const uint64_t kp = kprime_of(lo, hi, sh);
                    const uint64_t f = digit_reverse(kp, sh.weight), rv = (~kp) & sh.keymask;
                    sf[j] = rv < f; key[j] = sf[j] ? rv : f;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t gp = gp0 + j;
                key[j] = ~0ULL; sf[j] = 0;
                if (gp >= P) continue;
                const int g = genome_of(gp, tab);
                const uint32_t p = gp - tab.gpos_off[g];
                if (NARROW) {
                    const uint32_t kp = kprime_narrow(packed + tab.word_off[g], p, sh);
                    const uint32_t f = digit_reverse32(kp, sh.weight), rv = (~kp) & (uint32_t)sh.keymask;
                    sf[j] = rv < f; key[j] = sf[j] ? rv : f;
                } else {
                    const uint64_t kp = kprime_at(packed + tab.word_off[g], p, sh);
                    const uint64_t f = digit_reverse(kp, sh.weight), rv = (~kp) & sh.keymask;
                    sf[j] = rv < f; key[j] = sf[j] ? rv : f;
                }
            }
        }
        if (SEG || vmask || cmask) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t gp = gp0 + j;
                if (gp >= P) continue;
                const int g = together ? g0 : genome_of(gp, tab);
                const uint32_t p = gp - tab.gpos_off[g];
                if (SEG) {
                    const uint32_t *sg = seg + (size_t)g * (nseg + 1);
                    const uint32_t k = seg_of(sg, nseg, p);
                    key[j] = (p + sh.span <= sg[k + 1]) ? (((uint64_t)k << (2 * sh.weight)) | key[j]) : ~0ULL;
                }
                if ((vmask || cmask) && window_blocked(vmask ? vmask + tab.mask_off[g] : nullptr, cmask ? cmask + tab.mask_off[g] : nullptr, p, sh.span)) key[j] = ~0ULL;
            }
        }
        if (gp0 + 3 < P) {
            if (sizeof(KeyT) == 4) *reinterpret_cast<uint4 *>(keys + gp0) = make_uint4((uint32_t)key[0], (uint32_t)key[1], (uint32_t)key[2], (uint32_t)key[3]);
            else {
                reinterpret_cast<ulonglong2 *>(keys + gp0)[0] = make_ulonglong2(key[0], key[1]);
                reinterpret_cast<ulonglong2 *>(keys + gp0)[1] = make_ulonglong2(key[2], key[3]);
            }
            *reinterpret_cast<uint4 *>(vals + gp0) = make_uint4(gp0 | (sf[0] << 31), (gp0 + 1) | (sf[1] << 31), (gp0 + 2) | (sf[2] << 31), (gp0 + 3) | (sf[3] << 31));
#pragma unroll
            for (int j = 0; j < 4; j++) atomicAdd(&h[(uint32_t)((KeyT)key[j] >> hist_shift) & 255u], 1u);
        } else {
            for (int j = 0; j < 4 && gp0 + j < P; j++) {
                keys[gp0 + j] = (KeyT)key[j]; vals[gp0 + j] = (gp0 + j) | (sf[j] << 31);
                atomicAdd(&h[(uint32_t)((KeyT)key[j] >> hist_shift) & 255u], 1u);
            }
        }
    }
    __syncthreads();
    hist[threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

// ------------------------------------------------------------------------------------------------
// radix sort (LSD, 8-bit digits, stable): histogram / row scan / scatter per pass
// ------------------------------------------------------------------------------------------------
// block-wide exclusive scan of one value per thread (256 threads); returns the exclusive prefix, *total
// receives the block sum.  One global atomic per block instead of one per wave keeps a single output
// counter far below its ~12 ns-per-atomic serial rate (MI355X_MICROARCH.md "fanin").
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *total, uint32_t *lds /*[8]*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { uint32_t c = lds[w]; if (w < wave) wbase += c; tot += c; }
    __syncthreads();
    *total = tot;
    return wbase + inc - v;
}

// ------------------------------------------------------------------------------------------------
// Masked seed passes (guide-tree nodes below the root, LCB extension): most windows touch a masked base and
// would only be carried through the sort as dead all-ones keys.  Instead the valid windows are compacted, in
// position order (so equal mers still arrive in ascending position, as the join expects): count per tile,
// one-block scan over the tiles, then an extract that writes each tile's valid (key, val) pairs at the tile's
// offset.  A thread owns 16 consecutive windows, so one block scan of per-thread counts gives the order.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool window_valid(uint32_t gp, uint32_t P, const GenomeTab &tab, int span, const uint64_t *__restrict__ vmask,
                                             const uint64_t *__restrict__ cmask)
{
    if (gp >= P) return false;
    const int g = genome_of(gp, tab);
    return !window_blocked(vmask ? vmask + tab.mask_off[g] : nullptr, cmask ? cmask + tab.mask_off[g] : nullptr, gp - tab.gpos_off[g], span);
}

__global__ void __launch_bounds__(256) valid_count(GenomeTab tab, int span, uint32_t P, const uint64_t *__restrict__ vmask,
                                                   uint32_t *__restrict__ tile_cnt, const uint64_t *__restrict__ cmask)
{
    __shared__ uint32_t lds[8];
    const uint32_t first = blockIdx.x * 4096u + threadIdx.x * 16u;
    uint32_t c = 0;
    for (int i = 0; i < 16; i++) c += window_valid(first + i, P, tab, span, vmask, cmask) ? 1u : 0u;
    uint32_t total;
    (void)block_excl_scan(c, &total, lds);
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
}

// exclusive scan of tile_cnt[0..nblk) in place, total -> *total_out (one workgroup)
__global__ void __launch_bounds__(256) tile_scan(uint32_t *__restrict__ tile_cnt, uint32_t nblk, uint32_t *__restrict__ total_out)
{
    __shared__ uint32_t lds[8];
    const uint32_t chunk = (nblk + 255) / 256, lo = threadIdx.x * chunk, hi = min(lo + chunk, nblk);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += tile_cnt[i];
    uint32_t total;
    uint32_t run = block_excl_scan(sum, &total, lds);
    for (uint32_t i = lo; i < hi; i++) { const uint32_t v = tile_cnt[i]; tile_cnt[i] = run; run += v; }
    if (threadIdx.x == 0) *total_out = total;
}

template <typename KeyT, bool NARROW>
__global__ void __launch_bounds__(256) seed_extract_compact(const uint64_t *__restrict__ packed, GenomeTab tab, SeedShape sh,
                                                            KeyT *__restrict__ keys, uint32_t *__restrict__ vals, uint32_t P,
                                                            const uint64_t *__restrict__ vmask, const uint32_t *__restrict__ tile_off,
                                                            const uint64_t *__restrict__ cmask)
{
    __shared__ uint32_t lds[8];
    const uint32_t first = blockIdx.x * 4096u + threadIdx.x * 16u;
    uint32_t ok = 0;
    for (int i = 0; i < 16; i++) ok |= (window_valid(first + i, P, tab, sh.span, vmask, cmask) ? 1u : 0u) << i;
    uint32_t total;
    uint32_t o = tile_off[blockIdx.x] + block_excl_scan((uint32_t)__popc(ok), &total, lds);
    for (int i = 0; i < 16; i++) {
        if (!(ok >> i & 1)) continue;
        const uint32_t gp = first + i;
        const int g = genome_of(gp, tab);
        const uint32_t p = gp - tab.gpos_off[g];
        uint64_t key; uint32_t s;
        if (NARROW) {
            const uint32_t kp = kprime_narrow(packed + tab.word_off[g], p, sh);
            const uint32_t f = digit_reverse32(kp, sh.weight), r = (~kp) & (uint32_t)sh.keymask;
            s = r < f; key = s ? r : f;
        } else {
            const uint64_t kp = kprime_at(packed + tab.word_off[g], p, sh);
            const uint64_t f = digit_reverse(kp, sh.weight), r = (~kp) & sh.keymask;
            s = r < f; key = s ? r : f;
        }
        keys[o] = (KeyT)key; vals[o] = gp | (s << 31);
        o++;
    }
}

constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 16;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;   // 4096 keys per workgroup
constexpr int RS_WAVES = RS_THREADS / 64;

template <typename KeyT>
__global__ void __launch_bounds__(RS_THREADS) rs_hist(const KeyT *__restrict__ keys, uint32_t n, int shift,
                                                      uint32_t *__restrict__ hist, uint32_t nblk)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * RS_TILE;
    if (base + RS_TILE <= n) {
        // full tile: 16-byte loads, no predication (the order inside a tile does not matter for a histogram)
        constexpr int KPV = 16 / (int)sizeof(KeyT);
        struct alignas(16) Vec { KeyT k[KPV]; };
        const Vec *src = reinterpret_cast<const Vec *>(keys + base);
#pragma unroll
        for (int i = 0; i < RS_ITEMS / KPV; i++) {
            const Vec q = src[i * RS_THREADS + threadIdx.x];
#pragma unroll
            for (int j = 0; j < KPV; j++) atomicAdd(&h[(uint32_t)(q.k[j] >> shift) & 255u], 1u);
        }
    } else {
#pragma unroll
        for (int i = 0; i < RS_ITEMS; i++) {
            const uint32_t idx = base + i * RS_THREADS + threadIdx.x;
            if (idx < n) atomicAdd(&h[(uint32_t)(keys[idx] >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    hist[threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

// block d: exclusive scan of row d (length nblk) in place, row total -> totals[d]
// (a wave per quarter of the digit's row, 64 consecutive counts per step -- coalesced loads, a DPP prefix sum, one carry -- instead of a thread per 24
// consecutive counts walking them twice with loads 96 bytes apart: 17 -> 6 us per launch at C3's 6 104 tiles)
__device__ __forceinline__ uint32_t wave_incl_sum_u32(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)v, 0x111, 0xf, 0xf, false); v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)v, 0x114, 0xf, 0xf, false); v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)v, 0x142, 0xa, 0xf, false); v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)v, 0x143, 0xc, 0xf, false);
    return v;
}
__global__ void __launch_bounds__(256) rs_rowscan(uint32_t *__restrict__ hist, uint32_t nblk,
                                                  uint32_t *__restrict__ totals)
{
    __shared__ uint32_t part[4];                             // (launched with 256 threads: four waves, a quarter of the row each)
    uint32_t *row = hist + (size_t)blockIdx.x * nblk;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t seg = (((nblk + 3) / 4) + 63) & ~63u;           // counts per wave, whole steps of 64
    const uint32_t lo = min((uint32_t)wv * seg, nblk), hi = min(lo + seg, nblk);
    // (eight steps' loads in flight at a time: a loop that waits for one load per step is a chain of L2 latencies, 24 of them at C3)
    uint32_t s = 0;
    for (uint32_t i0 = lo; i0 < hi; i0 += 512) {
        uint32_t v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { const uint32_t idx = i0 + q * 64 + lane; v[q] = idx < hi ? row[idx] : 0u; }
#pragma unroll
        for (int q = 0; q < 8; q++) s += v[q];
    }
    s = wave_incl_sum_u32(s);
    if (lane == 63) part[wv] = s;
    __syncthreads();
    uint32_t run = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const uint32_t c = part[w]; if (w < wv) run += c; total += c; }
    for (uint32_t i0 = lo; i0 < hi; i0 += 512) {
        uint32_t v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { const uint32_t idx = i0 + q * 64 + lane; v[q] = idx < hi ? row[idx] : 0u; }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t idx = i0 + q * 64 + lane;
            const uint32_t inc = wave_incl_sum_u32(v[q]);
            if (idx < hi) row[idx] = run + inc - v[q];
            run += (uint32_t)__builtin_amdgcn_readlane((int32_t)inc, 63);
        }
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = total;
}

// One tile of the scatter.  FULL: the tile holds RS_TILE keys, so no lane is ever predicated off (all tiles but the
// last) -- the loads, ballots and stores compile without exec-mask branches.
template <typename KeyT, bool FULL>
__device__ __forceinline__ void rs_scatter_tile(const KeyT *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                KeyT *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                                uint32_t tile_base, uint32_t tile_n, int shift, KeyT *s_keys, uint32_t *s_vals,
                                                uint32_t (*wcount)[256], const uint32_t *gbase, uint32_t *tstart, uint32_t *scan)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // load (wave-striped: wave w owns [w*1024, (w+1)*1024), item i of lane l is index i*64+l)
    KeyT k[RS_ITEMS]; uint32_t v[RS_ITEMS]; uint32_t rank[RS_ITEMS];
    const uint32_t wbase = wave * (64 * RS_ITEMS);
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const uint32_t li = wbase + i * 64 + lane;
        const bool ok = FULL || li < tile_n;
        k[i] = ok ? keys_in[tile_base + li] : (KeyT)0;
        v[i] = ok ? vals_in[tile_base + li] : 0u;
    }
    // wave-level multisplit ranking, stable in (i, lane) order.  peers = lanes of row i with the same digit, built
    // as two 32-bit halves from eight ballots (one 3-input bit op per half and bit).  The wave owns
    // wcount[wave][]: every peer reads the running count, then the first peer bumps it -- LDS operations of one
    // wave stay in order, so no atomic and no broadcast is needed.
    uint32_t *wc = wcount[wave];
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const uint32_t li = wbase + i * 64 + lane;
        const bool ok = FULL || li < tile_n;
        const uint32_t d = (uint32_t)(k[i] >> shift) & 255u;
        uint32_t plo, phi;
        if (FULL) { plo = 0xffffffffu; phi = 0xffffffffu; }
        else { const uint64_t a = __ballot(ok); plo = (uint32_t)a; phi = (uint32_t)(a >> 32); }
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const uint64_t m = FULL ? __ballot((d >> b) & 1) : __ballot(ok && ((d >> b) & 1));
            const uint32_t sb = (uint32_t)((int32_t)(d << (31 - b)) >> 31);     // all ones when bit b of d is set
            plo &= ~((uint32_t)m ^ sb); phi &= ~((uint32_t)(m >> 32) ^ sb);
        }
        const uint32_t below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
        const uint32_t old = ok ? wc[d] : 0u;
        if (ok && below == 0) wc[d] = old + (uint32_t)__popc(plo) + (uint32_t)__popc(phi);
        rank[i] = old + below;
    }
    __syncthreads();
    // per-digit prefix over the waves, then the tile-level digit starts
    uint32_t c[RS_WAVES], sum = 0;
#pragma unroll
    for (int w = 0; w < RS_WAVES; w++) { c[w] = wcount[w][tid]; }
#pragma unroll
    for (int w = 0; w < RS_WAVES; w++) { uint32_t t = c[w]; wcount[w][tid] = sum; sum += t; }
    {
        uint32_t dummy;
        tstart[tid] = block_excl_scan(sum, &dummy, scan);
    }
    __syncthreads();
    // stage in LDS at the tile-sorted position
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const uint32_t li = wbase + i * 64 + lane;
        if (FULL || li < tile_n) {
            const uint32_t d = (uint32_t)(k[i] >> shift) & 255u;
            const uint32_t pos = tstart[d] + wc[d] + rank[i];
            s_keys[pos] = k[i]; s_vals[pos] = v[i];
        }
    }
    __syncthreads();
    // coalesced write-out: consecutive threads write consecutive addresses inside a digit's run
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const uint32_t pos = i * RS_THREADS + tid;
        if (FULL || pos < tile_n) {
            const KeyT kk = s_keys[pos];
            const uint32_t d = (uint32_t)(kk >> shift) & 255u;
            const uint32_t dst = gbase[d] + (pos - tstart[d]);
            keys_out[dst] = kk; vals_out[dst] = s_vals[pos];
        }
    }
}

// RAW: `hist` holds the raw per-tile digit counts (no rs_rowscan ran): every workgroup sums its digit's row for itself --
// the sorts of the chain / DP front / canonical order have at most a few dozen tiles, where the row scan is one more
// launch of pure latency.
template <typename KeyT, bool RAW = false>
__global__ void __launch_bounds__(RS_THREADS) rs_scatter(const KeyT *__restrict__ keys_in,
                                                         const uint32_t *__restrict__ vals_in,
                                                         KeyT *__restrict__ keys_out,
                                                         uint32_t *__restrict__ vals_out, uint32_t n, int shift,
                                                         const uint32_t *__restrict__ hist,
                                                         const uint32_t *__restrict__ totals, uint32_t nblk)
{
    __shared__ KeyT s_keys[RS_TILE];
    __shared__ uint32_t s_vals[RS_TILE];
    __shared__ uint32_t wcount[RS_WAVES][256];
    __shared__ uint32_t gbase[256];       // global output index of this tile's first key of digit d
    __shared__ uint32_t tstart[256];      // tile-local start of digit d
    __shared__ uint32_t scan[256];

    const int tid = threadIdx.x;
    const uint32_t tile_base = blockIdx.x * RS_TILE;
    const uint32_t tile_n = min((uint32_t)RS_TILE, n - tile_base);

    for (int w = 0; w < RS_WAVES; w++) wcount[w][tid] = 0;
    // digit base = exclusive scan of the digit totals
    if (RAW) {
        const uint32_t *row = hist + (size_t)tid * nblk;
        uint32_t tot = 0, before = 0;
        for (uint32_t t = 0; t < nblk; t++) { const uint32_t c = row[t]; tot += c; before += t < blockIdx.x ? c : 0u; }
        uint32_t dummy;
        gbase[tid] = block_excl_scan(tot, &dummy, scan) + before;
    } else {
        uint32_t dummy;
        const uint32_t tot = totals[tid];
        gbase[tid] = block_excl_scan(tot, &dummy, scan) + hist[(size_t)tid * nblk + blockIdx.x];
    }
    __syncthreads();
    if (tile_n == RS_TILE)
        rs_scatter_tile<KeyT, true>(keys_in, vals_in, keys_out, vals_out, tile_base, tile_n, shift, s_keys, s_vals, wcount, gbase, tstart, scan);
    else
        rs_scatter_tile<KeyT, false>(keys_in, vals_in, keys_out, vals_out, tile_base, tile_n, shift, s_keys, s_vals, wcount, gbase, tstart, scan);
}

// ------------------------------------------------------------------------------------------------
// mum_join: one thread per sorted entry; the thread at the first entry of a run of identical mers decides
// the hit by the finder's rule and scatters the hit into the dense HIT TABLE, indexed by the global
// window index of the hit's anchor (lowest genome of its component set):
//   tmask[p]    = component set (0 = no hit anchored at p)
//   tpos[p*N+g] = component g's window: global index | strand << 31   (one record of N words per anchor:
//                 a hit is written, and later read, as one contiguous record instead of N scattered words)
//   MODE_MEM    : MemHash -- a genome with more than one copy kills the seed
//   MODE_UNIQUE : UniqueMatchFinder.cpp:44-58 -- genomes with more than one copy are dropped, >= 2 stay
// A position carries at most one mer, so at most one hit is anchored at it: no atomics, no compaction.
// ------------------------------------------------------------------------------------------------
template <typename KeyT, bool SEG>
__global__ void __launch_bounds__(256) mum_join(const KeyT *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                uint32_t n, GenomeTab tab, int mode, uint32_t want_mask,
                                                uint32_t consider, uint32_t *__restrict__ tmask,
                                                uint32_t *__restrict__ tpos, uint32_t P, int has_invalid)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const KeyT k = keys[i];
    if ((SEG || has_invalid) && k == (KeyT)~0ULL) return;
    if (i > 0 && keys[i - 1] == k) return;
    if (i + 1 >= n || keys[i + 1] != k) return;        // singleton run
    // `consider` restricts the finder to a subset of the genomes (PairwiseMatchFinder: one pair at a time);
    // entries of the other genomes are invisible to it
    uint32_t once = 0, multi = 0, j = i;
    while (j < n && keys[j] == k) {
        uint32_t bit = (1u << genome_of(vals[j] & 0x7fffffffu, tab)) & consider;
        multi |= once & bit; once |= bit; j++;
        if (mode == MAUVE_MODE_MEM && multi) return;   // MemHash: a repeat kills the seed, no need to finish the run
    }
    const uint32_t m = once & ~multi;
    if (mode == MAUVE_MODE_MEM && multi) return;
    if (__popc(m) < 2) return;
    if (want_mask && m != want_mask) return;
    // entries of a run are in ascending global position (stable sort), so the anchor comes first
    uint32_t ap = 0xFFFFFFFFu;
    for (uint32_t t = i; t < j; t++) {
        const uint32_t v = vals[t], gp = v & 0x7fffffffu;
        const int g = genome_of(gp, tab);
        if (!(m >> g & 1)) continue;
        if (ap == 0xFFFFFFFFu) { ap = gp; tmask[ap] = m; }
        tpos[(size_t)ap * tab.nseq + g] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// join_hash: the finder rule without a full sort.  The (mer, position) pairs are radix-sorted on their HIGH bits
// only (bits [L, key_bits): as many 8-bit passes as bring the average bucket -- entries sharing those bits -- down to
// a few hundred); every occurrence of a mer then lies in one bucket, and a workgroup groups the mers of a range of
// whole buckets in an LDS hash table instead of ordering them: the finder rules only ask, per distinct mer, which
// genomes hold it once and which more than once -- the order of the mers is irrelevant, and the anchor (lowest genome
// of the component set) is found by genome id, not by position.  Two 8-bit passes and the join's read replace four
// passes, three histogram sweeps and a join that walked every run serially (C2: 30-bit mers, 15 M windows).
//
// Workgroup c owns [bound(c), bound(c+1)) -- the buckets that start in its chunk; bound(c) = first bucket boundary at
// or after c*HJ_T (chunk_bound; two waves look the two edges up side by side).  A range longer than the
// HJ_CAP entries its table takes (a bucket blown up by a repeat family or low-complexity sequence) is not processed:
// it goes to the overflow list and the host gives that slice to the full sort and the serial join below
// (rs_* + mum_join), which have no size limit.
// Slot = {mer (later the anchor's value), once | multi << 16 genome sets}; 4096 slots, at most 3072 entries (load <= 0.75).
// ------------------------------------------------------------------------------------------------
constexpr int HJ_T = 2048;                   // nominal entries per workgroup
constexpr int HJ_ROWS = 12;                  // rows of 256 entries a workgroup can take
constexpr int HJ_CAP = HJ_ROWS * 256;        // 3072
constexpr int HJ_SLOTS = 4096;
static_assert(HJ_CAP * 4 <= 3 * HJ_SLOTS, "join_hash: table load factor above 0.75");
constexpr uint32_t HJ_NONE = 0xffffffffu;
constexpr int HJ_OVF_CAP = 4096;            // ranges the overflow list holds; beyond that the whole list goes the old way

// first bucket boundary at or after chunk edge c (wave-uniform result): 64 entries per probe, a binary search (the
// list is ordered by the high bits) when a bucket is long
template <typename KeyT>
__device__ __forceinline__ uint32_t chunk_bound(const KeyT *__restrict__ keys, uint32_t n, int L, uint32_t c, int lane)
{
    const uint32_t start = c * (uint32_t)HJ_T;
    if (c == 0) return 0u;
    if (start >= n) return n;
    const uint64_t h0 = (uint64_t)keys[start - 1] >> L;                  // the bucket the edge falls into (or just behind)
    uint32_t res = HJ_NONE;
    for (int step = 0; step < 4 && res == HJ_NONE; step++) {              // 256 entries, 64 per probe
        const uint32_t idx = start + (uint32_t)step * 64u + (uint32_t)lane;
        const bool bnd = idx >= n || ((uint64_t)keys[idx] >> L) != h0;
        const uint64_t b = __ballot(bnd);
        if (b) res = start + (uint32_t)step * 64u + (uint32_t)(__ffsll((unsigned long long)b) - 1);
    }
    if (res == HJ_NONE) {
        // long bucket: first index in (lo, hi] whose high bits differ (keys[lo] still belongs to the bucket, index n counts)
        uint32_t lo = start + 255u, hi = n;
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (((uint64_t)keys[mid] >> L) != h0) hi = mid; else lo = mid;
        }
        res = hi;
    }
    return min(res, n);
}

template <typename KeyT>
__global__ void __launch_bounds__(256) join_bounds(const KeyT *__restrict__ keys, uint32_t n, int L, uint32_t nchunk,
                                                   uint32_t *__restrict__ bound)
{
    const uint32_t c = (blockIdx.x * 256u + threadIdx.x) >> 6;           // one wave per chunk edge 0 .. nchunk
    if (c > nchunk) return;
    const uint32_t res = chunk_bound(keys, n, L, c, threadIdx.x & 63);
    if ((threadIdx.x & 63) == 0) bound[c] = res;
}

template <typename KeyT, bool WIDE>
__global__ void __launch_bounds__(256) join_hash(const KeyT *__restrict__ keys, const uint32_t *__restrict__ vals, uint32_t n,
                                                 const uint32_t *__restrict__ bound, GenomeTab tab, int mode, uint32_t want_mask,
                                                 uint32_t *__restrict__ tmask, uint32_t *__restrict__ tpos,
                                                 uint32_t *__restrict__ ovf_cnt, uint32_t *__restrict__ ovf, uint32_t P)
{
    __shared__ KeyT skey[HJ_SLOTS];
    __shared__ uint32_t som[HJ_SLOTS];                       // once (low half) | multi (high half); WIDE: once only
    __shared__ uint32_t som2[WIDE ? HJ_SLOTS : 1];           // WIDE (> 16 genomes): multi
    constexpr KeyT EMPTY = (KeyT)~0ULL;                       // never a canonical mer; also the invalid-window key
    const int tid = threadIdx.x;
    const uint32_t cs = blockIdx.x * (uint32_t)HJ_T;
    // the rows sit at fixed places (cs + r*256 + tid), so their loads do not wait for the range: the first nine go out
    // at once; the range only decides which entries take part
    KeyT k[HJ_ROWS]; uint32_t v[HJ_ROWS];
#pragma unroll
    for (int r = 0; r <= HJ_T / 256; r++) {
        const uint32_t idx = cs + (uint32_t)r * 256u + (uint32_t)tid;
        k[r] = idx < n ? keys[idx] : EMPTY; v[r] = idx < n ? vals[idx] : 0u;
    }
    for (int i = tid; i < HJ_SLOTS; i += 256) { skey[i] = EMPTY; som[i] = 0; if (WIDE) som2[i] = 0; }
    __syncthreads();
    const uint32_t lo = bound[blockIdx.x], hi = bound[blockIdx.x + 1];
    if (lo >= hi) return;                                     // no bucket starts in this chunk
    if (hi > cs + (uint32_t)HJ_CAP) {                         // oversize: hand the range to the host
        if (tid == 0) {
            const uint32_t o = atomicAdd(&ovf_cnt[0], 1u);
            if (o < (uint32_t)HJ_OVF_CAP) { ovf[2 + 2 * o] = lo; ovf[3 + 2 * o] = hi; }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < HJ_ROWS; r++) {
        const uint32_t idx = cs + (uint32_t)r * 256u + (uint32_t)tid;
        if (r > HJ_T / 256) {                                 // rows the range rarely reaches
            k[r] = EMPTY; v[r] = 0u;
            if (cs + (uint32_t)r * 256u < hi && idx < hi) { k[r] = keys[idx]; v[r] = vals[idx]; }
        }
        if (idx < lo || idx >= hi) k[r] = EMPTY;
    }
    // ---- pass 1: group by mer: claim a slot (compare-and-swap on the key word, linear probing), then OR the entry's genome
    // into the slot's once / multi sets ----
    uint32_t slot[HJ_ROWS]; uint32_t gbit[HJ_ROWS];
#pragma unroll
    for (int r = 0; r < HJ_ROWS; r++) {
        slot[r] = HJ_NONE; gbit[r] = 0;
        if (k[r] == EMPTY) continue;                          // beyond the range, or an invalid window
        uint32_t s = ((uint32_t)k[r] ^ (uint32_t)((uint64_t)k[r] >> 32)) * 0x9E3779B1u >> 20;       // 12 bits
        for (;;) {
            const KeyT old = atomicCAS(&skey[s], EMPTY, k[r]);
            if (old == EMPTY || old == k[r]) break;
            s = (s + 1) & (HJ_SLOTS - 1);
        }
        const uint32_t bit = 1u << genome_of(v[r] & 0x7fffffffu, tab);
        const uint32_t old = atomicOr(&som[s], bit);
        if (old & bit) { if (WIDE) atomicOr(&som2[s], bit); else atomicOr(&som[s], bit << 16); }
        slot[r] = s; gbit[r] = bit;
    }
    __syncthreads();
    // ---- pass 2: the finder rule per mer; the entry of the lowest component genome is the anchor ----
    uint32_t mm[HJ_ROWS];
#pragma unroll
    for (int r = 0; r < HJ_ROWS; r++) {
        mm[r] = 0;
        if (slot[r] == HJ_NONE) continue;
        const uint32_t om = som[slot[r]];
        const uint32_t once = WIDE ? om : (om & 0xffffu), multi = WIDE ? som2[slot[r]] : (om >> 16);
        const uint32_t m = once & ~multi;
        if (mode == MAUVE_MODE_MEM && multi) continue;
        if (__popc(m) < 2 || (want_mask && m != want_mask) || !(m & gbit[r])) continue;
        mm[r] = m;
        if ((m & (0u - m)) == gbit[r]) skey[slot[r]] = (KeyT)v[r];          // grouping is over: the slot's mer makes room for the anchor
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < HJ_ROWS; r++) {
        if (!mm[r]) continue;
        const uint32_t ap = (uint32_t)skey[slot[r]] & 0x7fffffffu;
        if (ap >= P) { atomicAdd(&ovf_cnt[1], 1u); continue; }   // cannot happen; a wild store could take the device down
        tpos[(size_t)ap * tab.nseq + (__ffs(gbit[r]) - 1)] = v[r];
        if ((mm[r] & (0u - mm[r])) == gbit[r]) tmask[ap] = mm[r];
    }
}

// hits found by a host-side finder (mauve_extend_hits): record h = {component set, value of genome 0 .. N-1} -> hit table
// ---- tiny passes: extract + group + rule in ONE workgroup ------------------------------------------------------------------
// A seed pass over a few thousand windows (a round of the LCB extension over the pieces between the LCBs, a small guide-tree
// node) spends its time in launches: count, scan, extract, four sort passes, bounds, join -- a dozen kernels of 5-20 us each
// for a few kilobytes.  Here one workgroup of 1024 threads computes the canonical mers of ALL valid windows, groups them in an
// LDS hash table (compare-and-swap claim, linear probing; the same once / multi genome sets and the same finder rule as
// join_hash) and writes the hit table -- no key array, no sort.  <= TJ_MAX windows, <= 16 genomes, mer (+ gap id of a segmented
// set: the small batches of the deeper recursion levels) of <= 32 bits.
constexpr int TJ_SLOTS = 16384;                  // 128 KB of LDS: key word + genome sets per slot
constexpr int TJ_MAX = 9216;                     // load factor <= 0.5625
constexpr int TJ_ROWS = TJ_MAX / 1024;
__global__ void __launch_bounds__(1024) tiny_join(const uint64_t *__restrict__ packed, GenomeTab tab, SeedShape sh, uint32_t P,
                                                  const uint64_t *__restrict__ vmask, const uint64_t *__restrict__ cmask, int mode, uint32_t want_mask,
                                                  uint32_t *__restrict__ tmask, uint32_t *__restrict__ tpos, uint32_t *__restrict__ err_cnt,
                                                  const uint32_t *__restrict__ seg, uint32_t nseg, uint32_t *__restrict__ counters, uint32_t clr_lo, uint32_t clr_hi)
{
    extern __shared__ uint32_t tj_lds[];
    uint32_t *skey = tj_lds, *som = tj_lds + TJ_SLOTS;
    constexpr uint32_t EMPTY = 0xffffffffu;       // never a canonical mer (the smaller of a mer and its reverse complement)
    const int tid = threadIdx.x;
    // the pass's clears, here instead of two 5 us fill launches: the counter block and the slice of the hit table (the barriers below order them
    // before this workgroup's own stores to the table)
    if (counters && tid < 16) counters[tid] = 0;
    for (uint32_t i = clr_lo + tid; i < clr_hi; i += 1024) tmask[i] = 0;
    for (int i = tid; i < TJ_SLOTS; i += 1024) { skey[i] = EMPTY; som[i] = 0; }
    uint32_t v[TJ_ROWS], slot[TJ_ROWS], gbit[TJ_ROWS], key[TJ_ROWS];
#pragma unroll
    for (int r = 0; r < TJ_ROWS; r++) {
        const uint32_t gp = (uint32_t)r * 1024u + (uint32_t)tid;
        key[r] = EMPTY; v[r] = 0; gbit[r] = 0;
        if (!window_valid(gp, P, tab, sh.span, vmask, cmask)) continue;
        const int g = genome_of(gp, tab);
        const uint32_t p = gp - tab.gpos_off[g];
        uint32_t k, sflag;
        if (sh.span <= 32 && sh.weight <= 15) {
            const uint32_t kp = kprime_narrow(packed + tab.word_off[g], p, sh);
            const uint32_t f = digit_reverse32(kp, sh.weight), rr = (~kp) & (uint32_t)sh.keymask;
            sflag = rr < f; k = sflag ? rr : f;
        } else {
            const uint64_t kp = kprime_at(packed + tab.word_off[g], p, sh);
            const uint64_t f = digit_reverse(kp, sh.weight), rr = (~kp) & sh.keymask;
            sflag = rr < f; k = (uint32_t)(sflag ? rr : f);
        }
        if (seg) {                                  // segmented set (a batch of recursion gaps): the gap id rides above the mer, no window across gaps
            const uint32_t *sg = seg + (size_t)g * (nseg + 1);
            const uint32_t sk = seg_of(sg, nseg, p);
            if (p + (uint32_t)sh.span > sg[sk + 1]) continue;
            k |= sk << (2 * sh.weight);
        }
        key[r] = k; v[r] = gp | (sflag << 31); gbit[r] = 1u << g;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < TJ_ROWS; r++) {
        slot[r] = EMPTY;
        if (key[r] == EMPTY) continue;
        uint32_t sl = (key[r] * 0x9E3779B1u) >> 18;          // 14 bits
        for (;;) {
            const uint32_t old = atomicCAS(&skey[sl], EMPTY, key[r]);
            if (old == EMPTY || old == key[r]) break;
            sl = (sl + 1) & (TJ_SLOTS - 1);
        }
        const uint32_t old = atomicOr(&som[sl], gbit[r]);
        if (old & gbit[r]) atomicOr(&som[sl], gbit[r] << 16);
        slot[r] = sl;
    }
    __syncthreads();
    uint32_t mm[TJ_ROWS];
#pragma unroll
    for (int r = 0; r < TJ_ROWS; r++) {
        mm[r] = 0;
        if (slot[r] == EMPTY) continue;
        const uint32_t om = som[slot[r]], once = om & 0xffffu, multi = om >> 16, m = once & ~multi;
        if (mode == MAUVE_MODE_MEM && multi) continue;
        if (__popc(m) < 2 || (want_mask && m != want_mask) || !(m & gbit[r])) continue;
        mm[r] = m;
        if ((m & (0u - m)) == gbit[r]) skey[slot[r]] = v[r];          // grouping is over: the slot's mer makes room for the anchor
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < TJ_ROWS; r++) {
        if (!mm[r]) continue;
        const uint32_t ap = skey[slot[r]] & 0x7fffffffu;
        if (ap >= P) { atomicAdd(err_cnt, 1u); continue; }     // cannot happen (the anchor is one of the group's own windows)
        tpos[(size_t)ap * tab.nseq + (__ffs(gbit[r]) - 1)] = v[r];
        if ((mm[r] & (0u - mm[r])) == gbit[r]) tmask[ap] = mm[r];
    }
}

__global__ void __launch_bounds__(256) hits_scatter(const uint32_t *__restrict__ rec, uint32_t nh, int N, uint32_t P,
                                                    uint32_t *__restrict__ tmask, uint32_t *__restrict__ tpos)
{
    const uint32_t h = blockIdx.x * 256u + threadIdx.x;
    if (h >= nh) return;
    const uint32_t *r = rec + (size_t)h * (N + 1);
    const uint32_t m = r[0];
    const uint32_t ap = r[1 + (__ffs(m) - 1)] & 0x7fffffffu;
    if (ap >= P) return;                                      // validated on the host; never a wild store
    tmask[ap] = m;
    for (int g = 0; g < N; g++) if (m >> g & 1) tpos[(size_t)ap * N + g] = r[1 + g];
}

// ------------------------------------------------------------------------------------------------
// PairwiseMatchFinder: N(N-1)/2 finder passes over ONE sorted mer list.  Instead of re-reading all P sorted entries
// per pair, one pass (run_summary) lists the runs of identical mers that could matter to any pair -- where the run
// starts, how long it is, which genomes occur in it exactly once -- and each pair's join (join_pair) walks that
// list: a run is a hit of pair (i, j) iff both genomes are in its exactly-once set (MemHash restricted to the two
// genomes: entries of the others are invisible to it).  The order of the list does not matter: hits go to the
// dense table by anchor position.
// ------------------------------------------------------------------------------------------------
template <typename KeyT>
__global__ void __launch_bounds__(256) run_summary(const KeyT *__restrict__ keys, const uint32_t *__restrict__ vals, uint32_t n,
                                                   GenomeTab tab, int has_invalid, uint32_t *__restrict__ rstart,
                                                   uint32_t *__restrict__ rlen, uint32_t *__restrict__ runiq,
                                                   uint32_t *__restrict__ counter, uint32_t cap)
{
    // 1024 entries per workgroup (four per thread, strided), one block scan and ONE global atomic per workgroup: with one atomic per
    // wave, 250 k of them on the same counter were the kernel's whole run time (2.9 ms for 16 M entries)
    __shared__ uint32_t lds[8];
    __shared__ uint32_t s_base;
    constexpr int ITEMS = 4;
    const uint32_t base = blockIdx.x * (256u * ITEMS);
    uint32_t uniq[ITEMS], len[ITEMS], cnt = 0;
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const uint32_t i = base + it * 256 + threadIdx.x;
        uniq[it] = 0; len[it] = 0;
        if (i < n) {
            const KeyT k = keys[i];
            const bool valid = !(has_invalid && k == (KeyT)~0ULL);
            if (valid && (i == 0 || keys[i - 1] != k) && i + 1 < n && keys[i + 1] == k) {
                uint32_t once = 0, multi = 0, j = i;
                while (j < n && keys[j] == k) {
                    const uint32_t bit = 1u << genome_of(vals[j] & 0x7fffffffu, tab);
                    multi |= once & bit; once |= bit; j++;
                }
                uniq[it] = once & ~multi; len[it] = j - i;
            }
        }
        if (__popc(uniq[it]) >= 2) cnt++; else uniq[it] = 0;
    }
    uint32_t total;
    const uint32_t off = block_excl_scan(cnt, &total, lds);
    if (threadIdx.x == 0) s_base = total ? atomicAdd(counter, total) : 0u;
    __syncthreads();
    uint32_t r = s_base + off;
#pragma unroll
    for (int it = 0; it < ITEMS; it++)
        if (uniq[it]) { if (r < cap) { rstart[r] = base + it * 256 + threadIdx.x; rlen[r] = len[it]; runiq[r] = uniq[it]; } r++; }   // (the counter keeps counting: the host sees a list that outgrew its buffer)
}

__global__ void __launch_bounds__(256) join_pair(const uint32_t *__restrict__ vals, GenomeTab tab, const uint32_t *__restrict__ rstart,
                                                 const uint32_t *__restrict__ rlen, const uint32_t *__restrict__ runiq,
                                                 uint32_t nruns, int gi, int gj, uint32_t *__restrict__ tmask,
                                                 uint32_t *__restrict__ tpos)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nruns) return;
    const uint32_t u = runiq[r];
    if (!((u >> gi) & (u >> gj) & 1u)) return;
    const uint32_t s = rstart[r], L = rlen[r];
    uint32_t vi = 0, vj = 0;
    for (uint32_t t = s; t < s + L; t++) {
        const uint32_t v = vals[t];
        const int g = genome_of(v & 0x7fffffffu, tab);
        if (g == gi) vi = v;
        if (g == gj) vj = v;
    }
    const uint32_t ap = vi & 0x7fffffffu;               // gi < gj: the anchor is genome gi's window
    tmask[ap] = (1u << gi) | (1u << gj);
    tpos[(size_t)ap * tab.nseq + gi] = vi;
    tpos[(size_t)ap * tab.nseq + gj] = vj;
}

// ------------------------------------------------------------------------------------------------
// SeedMatchEnumerator on the device (SeedMatchEnumerator.h:59-141): the sorted mer list of ONE sequence is cut into
// runs of identical mers; a run of m occurrences, min_multi <= m <= max_multi, becomes one match of m components in
// position order (the stable sort keeps them so), 1-based, a component negative when its strand flag differs from
// the first component's (SetDirection, :127-141); with direct_only a run that has reverse components keeps only the
// forward ones, if two or more remain (:88-117).  Output CSR: mult[r], start_off[r], starts[].
//   enum_runs  : the head of every run walks it and notes how many starts it will emit (0: no match)
//   vscan_*    : start offsets; cmp_* (EnumRuns): the emitting runs in list order -> mult / start_off
//   enum_write : the heads walk again and write the starts
// ------------------------------------------------------------------------------------------------
template <typename KeyT>
__global__ void __launch_bounds__(256) enum_runs(const KeyT *__restrict__ keys, const uint32_t *__restrict__ vals, uint32_t n, int64_t min_multi,
                                                 int64_t max_multi, int direct_only, uint32_t *__restrict__ emit)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const KeyT k = keys[i];
    uint32_t e = 0;
    if (i == 0 || keys[i - 1] != k) {
        const uint32_t ref = vals[i] >> 31;
        uint32_t j = i, kept = 0; bool found_rev = false;
        while (j < n && keys[j] == k) { if ((vals[j] >> 31) != ref) found_rev = true; else kept++; j++; }
        const int64_t m = (int64_t)(j - i);
        if (m >= 2 && m >= min_multi && m <= max_multi) e = (direct_only && found_rev) ? (kept > 1 ? kept : 0u) : (uint32_t)m;
    }
    emit[i] = e;
}
struct EmitVal { const uint32_t *e; __device__ int64_t value(uint32_t i) const { return e[i]; } };
struct EnumRuns {
    const uint32_t *cnt; const int64_t *soff; uint32_t n; int64_t *mult, *start_off; int64_t *tot /* [0] runs, [1] starts */;
    __device__ uint32_t domain(int) const { return n; }
    __device__ bool flag(uint32_t i, int) const { return cnt[i] != 0; }
    __device__ void each(uint32_t, uint32_t, bool, int) const {}
    __device__ void emit(uint32_t i, uint32_t r, int) const { mult[r] = cnt[i]; start_off[r] = soff[i]; }
    __device__ void total(uint32_t runs, int) const { tot[0] = runs; start_off[runs] = soff[n]; }
};
template <typename KeyT>
__global__ void __launch_bounds__(256) enum_write(const KeyT *__restrict__ keys, const uint32_t *__restrict__ vals, uint32_t n, int direct_only,
                                                  const uint32_t *__restrict__ emit, const int64_t *__restrict__ soff, uint32_t gpos0,
                                                  int64_t *__restrict__ starts)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n || emit[i] == 0) return;
    const KeyT k = keys[i];
    const uint32_t ref = vals[i] >> 31;
    uint32_t j = i; bool found_rev = false;
    while (j < n && keys[j] == k) { if ((vals[j] >> 31) != ref) found_rev = true; j++; }
    int64_t o = soff[i];
    for (uint32_t t = i; t < j; t++) {
        const bool rv = (vals[t] >> 31) != ref;
        const int64_t p1 = (int64_t)((vals[t] & 0x7fffffffu) - gpos0) + 1;
        if (direct_only && found_rev) { if (!rv) starts[o++] = p1; }
        else starts[o++] = rv ? -p1 : p1;
    }
}

// ------------------------------------------------------------------------------------------------
// extension.  A hit fixes a generalized diagonal: component c moves +k (same strand as the
// anchor) or -k (opposite strand) when the anchor moves +k.  Offset k "agrees" when the masked windows of
// all components are equal there.  Agreeing offsets at most `span` apart chain into a cluster; the match
// is the cluster through the hit, emitted by its leftmost same-mask hit.
// ------------------------------------------------------------------------------------------------
// One component of the hit a wave is extending: where its windows live, the window index at offset 0, its walking
// direction relative to the anchor and the window range it may use.  Built once per candidate (LDS, per wave).
struct ExtComp { const uint64_t *G; const uint64_t *VM; const uint64_t *CM; int64_t pos, lo, hi; uint32_t rev, maxw /* last word of the genome's buffer */, g, pad; };

// ---- does offset k agree? ----
// Offset k agrees when, for every component c and every care position t of the seed, S_c(k + t) == S_0(k + t), where S_c(j) is the
// component's base stream along the generalized diagonal: base(pos_c + j) for a component on the anchor's strand, the COMPLEMENT of
// base(pos_c + span - 1 - j) for one on the opposite strand (its window at pos_c - k against the reverse complement of the anchor's; the
// pattern is a palindrome, so t and span - 1 - t are care positions together).  A round of 64 offsets from j0 on needs S(j0 .. j0 + 63 +
// span - 1): 128 digits, four words per component.  The wave builds them with its lanes side by side -- eight lanes per component, lane i
// of a group fetches word i of the stretch (ONE load instruction for up to eight components) -- XORs each with the anchor's, ORs the
// differences over the components and hands the four words out as wave-uniform values; a lane then cuts its window out of them and tests
// it under the care mask.  (Every lane fetching its own window, three words per component, and a version with the streams built by the
// scalar unit, both spent their time issuing instructions: 800 and 550 per round against about 200.)
__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src)
{
    return (uint64_t)(uint32_t)__shfl((int)(uint32_t)v, src) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), src) << 32);
}
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l)
{
    return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int32_t)(uint32_t)v, l) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int32_t)(uint32_t)(v >> 32), l) << 32);
}
// lane l <- lane l + 1 inside its row of 16 (the groups of eight never need the word of the next row)
__device__ __forceinline__ uint64_t row_next64(uint64_t v)
{
    return (uint64_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)(uint32_t)v, 0x101, 0xf, 0xf, true) |
           ((uint64_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)(uint32_t)(v >> 32), 0x101, 0xf, 0xf, true) << 32);
}
// M: OR over the components of S_c ^ S_0 for j = j0 .. j0 + 127 (two bits per digit), wave-uniform.  NJ stretches at once (the first left and
// the first right round of a candidate: their loads travel together).
template <int NJ>
__device__ __forceinline__ void ext_mismatch(const ExtComp *__restrict__ comp, int nc, int span, const int64_t (&j0)[NJ], int lane, uint64_t (&M)[NJ][4])
{
    typedef const uint64_t __attribute__((address_space(1))) *gptr;
    const int i = lane & 7, grp = lane >> 3;
    uint64_t acc[NJ], A[NJ];
#pragma unroll
    for (int t = 0; t < NJ; t++) { acc[t] = 0; A[t] = 0; }
    for (int c0 = 0; c0 < nc; c0 += 8) {                   // (wave-uniform: more than eight components only with more than eight genomes)
        const int c = c0 + grp;
        const ExtComp &C = comp[min(c, nc - 1)];
        const uint32_t rev = C.rev;
        const bool any_rev = __any(rev != 0);
        const gptr G = (gptr)(uintptr_t)C.G;
        const int64_t pos = C.pos, maxw = (int64_t)C.maxw;
        uint64_t w[NJ]; int r[NJ];
#pragma unroll
        for (int t = 0; t < NJ; t++) {
            // opposite strand: the forward stretch F that ends at base top = pos + span - 1 - j0; S digit k = ~F digit (127 - k)
            const int64_t b = rev ? pos + span - 1 - j0[t] - 127 : pos + j0[t];
            const int64_t q = b >> 5; r[t] = (int)(b & 31) * 2;
            // (b may lie outside the genome: word indices are clamped into its buffer; what they hold there only reaches offsets the bounds
            // test rejects anyway)
            int64_t x = q + i; x = x < 0 ? 0 : (x > maxw ? maxw : x);
            w[t] = G[x];
        }
#pragma unroll
        for (int t = 0; t < NJ; t++) {
            const uint64_t wn = row_next64(w[t]);          // (lanes i >= 4 compute along, unused)
            const uint64_t W = r[t] ? ((w[t] >> r[t]) | (wn << (64 - r[t]))) : w[t];
            uint64_t S = W;
            if (any_rev) { const uint64_t Wr = ~digit_reverse64(shfl64(W, (lane & ~7) | (3 - (i & 3)))); S = rev ? Wr : W; }
            if (c0 == 0) A[t] = shfl64(S, i & 3);          // the anchor (component 0, never reversed) word i, in every group
            if (c > 0 && c < nc && i < 4) acc[t] |= S ^ A[t];
        }
    }
#pragma unroll
    for (int t = 0; t < NJ; t++) {
        uint64_t v = acc[t];
        v |= shfl64(v, lane ^ 8); v |= shfl64(v, lane ^ 16); v |= shfl64(v, lane ^ 32);      // over the groups
#pragma unroll
        for (int k = 0; k < 4; k++) M[t][k] = readlane64(v, k);
    }
}
// The walk over one round's agreement bitmap A (bit i: offset i of the round agrees), from bit 0: jump to the next agreeing offset at most
// `span` away and over its run, until `span` offsets in a row disagree (done) or the bitmap cannot tell any more (the caller starts a fresh
// round at the returned position).  Closed form of that loop: the walk stops at the first run of >= span zeros that the 64 bits show whole,
// else at the first end of a run of ones beyond bit 64 - span.  Returns the offsets consumed; every agreeing offset below it was visited.
__device__ __forceinline__ int walk_round(uint64_t A, int span, bool &done)
{
    uint64_t R = ~A; int len = 1;
    while (2 * len <= span) { R &= R >> len; len *= 2; }
    if (len < span) R &= R >> (span - len);                // R bit i: offsets i .. i + span - 1 all disagree (zeros shifted in: never across bit 63)
    if (R) { done = true; return __ffsll((unsigned long long)R) - 1; }
    const uint64_t ends = A & ~(A >> 1) & (~0ULL << (64 - span));     // last offsets of the runs that reach beyond 64 - span (not empty when R is)
    return __ffsll((unsigned long long)ends);
}
// the window of the lane's offset: digits d .. d + span - 1 of M (0 <= d < 64), under the care mask
__device__ __forceinline__ bool ext_window_clean(const uint64_t M[4], int d, const SeedShape &sh)
{
    const bool up = d >= 32; const int r = (d & 31) * 2;
    const uint64_t a = up ? M[1] : M[0], b = up ? M[2] : M[1], c = up ? M[3] : M[2];
    const uint64_t lo = r ? ((a >> r) | (b << (64 - r))) : a, hi = r ? ((b >> r) | (c << (64 - r))) : b;
    return ((lo & sh.care_lo) | (hi & sh.care_hi)) == 0;
}
// placed / ambiguous bases and contig starts under the windows of offset k (only when the pass has such bitmaps)
__device__ __forceinline__ bool ext_blocked(const ExtComp *__restrict__ comp, int nc, int span, int64_t k)
{
    bool b = false;
    for (int c = 0; c < nc; c++) {
        const ExtComp &C = comp[c];
        const int64_t q = C.rev ? C.pos - k : C.pos + k;
        if ((C.VM || C.CM) && q >= C.lo && q <= C.hi) b |= window_blocked(C.VM, C.CM, (uint32_t)q, span);
    }
    return b;
}

// phase A (streaming, no genome access): one thread per window position.  A hit anchored at p that has a
// same-mask hit on the same diagonal at p-d, d <= span, cannot be the leftmost hit of its cluster (the
// left walk of S4 never jumps over an agreeing offset, and that hit agrees).  Everything else is a
// candidate for phase B.  all != 0 (no extension): every hit is a candidate.
constexpr int RUNS_ITEMS = 16;
constexpr uint32_t RUNS_TILE = 256u * RUNS_ITEMS, RUNS_HALO = 64 /* >= MAUVE_MAX_SEED_SPAN */;
static_assert(RUNS_HALO >= MAUVE_MAX_SEED_SPAN && RUNS_HALO == 64, "mum_runs looks back at most one span, and reads the mask word 64 positions in front of every lane");
static_assert(RUNS_ITEMS <= 32, "one bit per item in the hit / pending / flag words");

// same generalized diagonal: every component of hit p sits d windows after (forward) / before (reverse) the
// corresponding component of hit q = p - d, with the same relative strands
__device__ __forceinline__ bool same_diagonal(const uint32_t *__restrict__ rp, const uint32_t *__restrict__ rq, uint32_t m, int a,
                                              int nseq, uint32_t d)
{
    const uint32_t sa = rp[a] >> 31, sq = rq[a] >> 31;
    for (int g = a + 1; g < nseq; g++) {
        if (!(m >> g & 1)) continue;
        const uint32_t vp = rp[g], vq = rq[g];
        const uint32_t o = (vp >> 31) ^ sa;
        if (((vq >> 31) ^ sq) != o) return false;
        const uint32_t pp = vp & 0x7fffffffu, pq = vq & 0x7fffffffu;
        if (!(o ? (pq == pp + d) : (pq + d == pp))) return false;
    }
    return true;
}

template <bool SEG>
__device__ __forceinline__ void mum_runs_body(const GenomeTab &tab, int span, const uint32_t *__restrict__ tmask,
                                              const uint32_t *__restrict__ tpos, uint32_t P, int all,
                                              uint32_t *__restrict__ cand, uint32_t *__restrict__ counters,
                                              const uint32_t *__restrict__ seg, uint32_t nseg, uint32_t p0, uint32_t cand_cap)
{
    __shared__ uint32_t lds[8];
    __shared__ uint32_t s_base;
    __shared__ uint32_t s_mask[RUNS_HALO + RUNS_TILE];
    // RUNS_TILE windows per workgroup: one block scan and ONE global atomic per tile.  [p0, P): the positions looked at (a pass of
    // the pairwise finder only has hits in its lower genome)
    const uint32_t base = p0 + blockIdx.x * RUNS_TILE;
    const int N = tab.nseq;
    uint32_t flags = 0, cnt = 0;
    uint32_t mm[RUNS_ITEMS];
    uint32_t hits = 0;
#pragma unroll
    for (int it = 0; it < RUNS_ITEMS; it++) {
        const uint32_t p = base + it * 256 + threadIdx.x;
        mm[it] = p < P ? tmask[p] : 0u;
        if (mm[it]) hits |= 1u << it;
    }
    if (all) { flags = hits; cnt = __popc(hits); }
    else if (__syncthreads_or(hits != 0)) {
        // Three phases, each with all its memory operations in flight at once (a hit-by-hit walk with dependent loads is what this kernel used
        // to spend its time in): (1) the mask words of the tile (and RUNS_HALO words in front of it) in LDS: every hit finds the nearest
        // earlier position with the same mask inside its genome / gap segment, at most span away -- none: a candidate; (2) the records of the
        // hit and of that position, all items together: on the same diagonal, the hit is not the leftmost of its cluster; (3) the rare hit
        // whose nearest same-mask neighbour lies on ANOTHER diagonal walks on from there, one offset at a time.
#pragma unroll
        for (int it = 0; it < RUNS_ITEMS; it++) s_mask[RUNS_HALO + it * 256 + threadIdx.x] = mm[it];
        if (threadIdx.x < RUNS_HALO) {
            const int64_t q = (int64_t)base - RUNS_HALO + threadIdx.x;
            s_mask[threadIdx.x] = (q >= 0 && q < (int64_t)P) ? tmask[q] : 0u;
        }
        __syncthreads();
        // (phases 1 and 2 are written without per-lane branches: the exec-mask bookkeeping of sixteen unrolled items was what the scalar unit
        // -- one per compute unit -- spent the kernel on)
        const int lane = threadIdx.x & 63;
        uint32_t dd[RUNS_ITEMS];
        uint32_t pend = 0, any = 0;
#pragma unroll
        for (int it = 0; it < RUNS_ITEMS; it++) {
            dd[it] = 0;
            const uint32_t m = mm[it];
            uint64_t left = __ballot(m != 0);
            if (left == 0) continue;                          // (wave-uniform)
            const uint32_t p = base + it * 256 + threadIdx.x;
            const int a = m ? __ffs(m) - 1 : 0;
            uint32_t g0 = tab.gpos_off[a];
            if (SEG) { if (m) { const uint32_t *sg = seg + (size_t)a * (nseg + 1); g0 += sg[seg_of(sg, nseg, p - g0)]; } }
            const uint32_t lim = m ? min((uint32_t)span, p - g0) : 0u;
            // the nearest earlier position with the same mask, from ballots: one pair per distinct mask among the wave's hits (few), the 64
            // positions in front of a lane as one word with p - 1 at bit 63
            const uint32_t mprev = s_mask[RUNS_HALO + it * 256 + threadIdx.x - 64];
            uint32_t d = 0;
            while (left) {
                const uint32_t mv = (uint32_t)__builtin_amdgcn_readlane((int32_t)m, __ffsll((unsigned long long)left) - 1);
                const uint64_t Bc = __ballot(m == mv), Bp = __ballot(mprev == mv);
                left &= ~Bc;
                const uint64_t W = lane ? ((Bp >> lane) | (Bc << (64 - lane))) : Bp;
                const uint32_t dist = W ? (uint32_t)__clzll((long long)W) + 1u : 0u;
                d = m == mv ? dist : d;
            }
            const bool pd = d != 0 && d <= lim, cd = m != 0 && !pd;      // has such a neighbour / is a candidate already
            dd[it] = pd ? d : 0u;
            pend |= (pd ? 1u : 0u) << it; any |= pd ? m : 0u;
            flags |= (cd ? 1u : 0u) << it; cnt += cd ? 1u : 0u;
        }
        uint32_t sa_bits = 0, sq_bits = 0, bad = 0;
        // (32-bit offsets from a wave-uniform base keep the sixteen address pairs out of the register file)
        const uint32_t base_h = base >= RUNS_HALO ? base - RUNS_HALO : 0u;
        const uint32_t *tb = tpos + (size_t)base_h * N;
        constexpr int RB = 8;                                 // items per batch of loads
#pragma unroll
        for (int i0 = 0; i0 < RUNS_ITEMS; i0 += RB) {
            if (!__any((pend >> i0) & ((1u << RB) - 1u))) continue;
            for (int g = 0; g < N; g++) {
                if (!__any((any >> g) & 1u)) continue;        // (wave-uniform: no pending hit of this wave has a component in genome g)
                uint32_t vp[RB], vq[RB];
#pragma unroll
                for (int i = 0; i < RB; i++) {                // (a lane with nothing to ask reads word 0 of the stretch: no branch)
                    const int it = i0 + i;
                    const uint32_t po = base - base_h + it * 256 + threadIdx.x;
                    const bool mine = ((pend >> it) & 1u) && ((mm[it] >> g) & 1u);
                    vp[i] = tb[mine ? po * (uint32_t)N + (uint32_t)g : 0u];
                    vq[i] = tb[mine ? (po - dd[it]) * (uint32_t)N + (uint32_t)g : 0u];
                }
#pragma unroll
                for (int i = 0; i < RB; i++) {
                    const int it = i0 + i;
                    const bool mine = ((pend >> it) & 1u) && ((mm[it] >> g) & 1u);
                    const bool first = g == __ffs(mm[it]) - 1;             // the anchor's component: its strands are the reference
                    sa_bits |= (mine && first) ? (vp[i] >> 31) << it : 0u; sq_bits |= (mine && first) ? (vq[i] >> 31) << it : 0u;
                    // same_diagonal(p, p - d), component g
                    const uint32_t o = (vp[i] >> 31) ^ ((sa_bits >> it) & 1u);
                    const uint32_t pp = vp[i] & 0x7fffffffu, pq = vq[i] & 0x7fffffffu;
                    const bool off = (((vq[i] >> 31) ^ ((sq_bits >> it) & 1u)) != o) || !(o ? (pq == pp + dd[it]) : (pq + dd[it] == pp));
                    bad |= (mine && !first && off) ? 1u << it : 0u;
                }
            }
        }
        const uint32_t again = pend & bad;
#pragma unroll
        for (int it = 0; it < RUNS_ITEMS; it++) {             // (unrolled: a run-time index would put mm / dd into scratch memory)
            if (!((again >> it) & 1u)) continue;
            const uint32_t p = base + it * 256 + threadIdx.x, m = mm[it];
            const int a = __ffs(m) - 1;
            uint32_t g0 = tab.gpos_off[a];
            if (SEG) { const uint32_t *sg = seg + (size_t)a * (nseg + 1); g0 += sg[seg_of(sg, nseg, p - g0)]; }
            const uint32_t *sm = s_mask + RUNS_HALO + it * 256 + threadIdx.x;
            const uint32_t lim = min((uint32_t)span, p - g0);
            bool is_cand = true;
            for (uint32_t d = dd[it] + 1; d <= lim && is_cand; d++) {
                if (sm[-(int)d] != m) continue;
                if (same_diagonal(tpos + (size_t)p * N, tpos + (size_t)(p - d) * N, m, a, N, d)) is_cand = false;
            }
            if (is_cand) { flags |= 1u << it; cnt++; }
        }
    }
    uint32_t total;
    const uint32_t off = block_excl_scan(cnt, &total, lds);
    if (threadIdx.x == 0) s_base = total ? atomicAdd(&counters[1], total) : 0u;
    __syncthreads();
    uint32_t o = s_base + off;
#pragma unroll
    for (int it = 0; it < RUNS_ITEMS; it++)
        if (flags >> it & 1) { if (o < cand_cap) cand[o] = base + it * 256 + threadIdx.x; o++; }     // never past the list: counters[1] still counts, the host refuses a list that outgrew its buffer
}
template <bool SEG>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) mum_runs(GenomeTab tab, int span, const uint32_t *__restrict__ tmask,
                                                const uint32_t *__restrict__ tpos, uint32_t P, int all,
                                                uint32_t *__restrict__ cand, uint32_t *__restrict__ counters,
                                                const uint32_t *__restrict__ seg, uint32_t nseg, uint32_t cand_cap, uint32_t p0 = 0)
{
    mum_runs_body<SEG>(tab, span, tmask, tpos, P, all, cand, counters, seg, nseg, p0, cand_cap);
}
// Several passes of the pairwise finder at once: pairs with DIFFERENT lower genomes write disjoint slices of the hit table, so a group of
// them shares the table, one candidate list and one counter (mum_extend tells the pairs apart by the hit's mask).  blockIdx.y = pair.
struct PairGroup { int n; int ga[MAUVE_MAX_SEQ], gb[MAUVE_MAX_SEQ]; uint32_t lo[MAUVE_MAX_SEQ], hi[MAUVE_MAX_SEQ]; };
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) mum_runs_group(GenomeTab tab, int span, const uint32_t *__restrict__ tmask, const uint32_t *__restrict__ tpos, PairGroup grp, int all,
                                                      uint32_t *__restrict__ cand, uint32_t *__restrict__ counters, uint32_t cand_cap)
{
    const uint32_t lo = grp.lo[blockIdx.y], hi = grp.hi[blockIdx.y];
    if (lo + blockIdx.x * RUNS_TILE >= hi) return;                     // (workgroup-uniform: the grid covers the longest slice)
    mum_runs_body<false>(tab, span, tmask, tpos, hi, all, cand, counters, nullptr, 0u, lo, cand_cap);
}
__global__ void __launch_bounds__(256) join_pair_group(const uint32_t *__restrict__ vals, GenomeTab tab, const uint32_t *__restrict__ rstart,
                                                       const uint32_t *__restrict__ rlen, const uint32_t *__restrict__ runiq,
                                                       uint32_t nruns, PairGroup grp, uint32_t *__restrict__ tmask, uint32_t *__restrict__ tpos)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nruns) return;
    const uint32_t u = runiq[r];
    uint32_t hit = 0;                                   // the pairs of the group this run is a hit of
    for (int y = 0; y < grp.n; y++) if ((u >> grp.ga[y]) & (u >> grp.gb[y]) & 1u) hit |= 1u << y;
    if (!hit) return;
    const uint32_t s = rstart[r], L = rlen[r];
    // the run's entries once (nearly every run has at most one entry per genome: <= 8 of them stay in registers)
    constexpr int RC = 8;
    uint32_t ev[RC]; int eg[RC];
#pragma unroll
    for (int t = 0; t < RC; t++) {
        ev[t] = 0; eg[t] = -1;
        if ((uint32_t)t < L) { ev[t] = vals[s + t]; eg[t] = genome_of(ev[t] & 0x7fffffffu, tab); }
    }
    for (; hit; hit &= hit - 1) {
        const int y = __ffs(hit) - 1, gi = grp.ga[y], gj = grp.gb[y];
        uint32_t vi = 0, vj = 0;
#pragma unroll
        for (int t = 0; t < RC; t++) { if (eg[t] == gi) vi = ev[t]; if (eg[t] == gj) vj = ev[t]; }
        for (uint32_t t = s + RC; t < s + L; t++) {     // (longer runs: the rest from memory)
            const uint32_t v = vals[t];
            const int g = genome_of(v & 0x7fffffffu, tab);
            if (g == gi) vi = v;
            if (g == gj) vj = v;
        }
        const uint32_t ap = vi & 0x7fffffffu;           // gi < gj: the anchor is genome gi's window
        tmask[ap] = (1u << gi) | (1u << gj);
        tpos[(size_t)ap * tab.nseq + gi] = vi;
        tpos[(size_t)ap * tab.nseq + gj] = vj;
    }
}

// phase B: one wave per candidate; the 64 lanes test 64 consecutive offsets at a time and the resulting
// agreement bitmap is walked with scalar bit operations.  Record slot = candidate index; length 0 marks a
// candidate that turned out not to be the leftmost hit of its cluster.
#ifdef MAUVE_EXT_STATS
// measurement build only (-DMAUVE_EXT_STATS): [0] rounds, [2] sum of wave times, [3] longest wave time (10 ns ticks), [4] waves, [5] most rounds of one wave, [1] / [6] / [7] time in set-up / left walk / right walk
__device__ unsigned long long g_ext_stats[8];
#endif
template <bool SEG>
__global__ void __launch_bounds__(256) mum_extend(const uint64_t *__restrict__ packed, GenomeTab tab, SeedShape sh,
                                                  const uint32_t *__restrict__ tmask, const uint32_t *__restrict__ tpos,
                                                  uint32_t P, const uint32_t *__restrict__ cand, uint32_t ncand,
                                                  int extend, int32_t *__restrict__ mlen, int32_t *__restrict__ mstart,
                                                  const uint32_t *__restrict__ seg, uint32_t nseg,
                                                  const uint64_t *__restrict__ vmask, const uint64_t *__restrict__ cmask,
                                                  const uint32_t *__restrict__ ncand_dev = nullptr, uint32_t *__restrict__ counters_out = nullptr)
{
    __shared__ ExtComp s_comp[4][MAUVE_MAX_SEQ];
    // ncand_dev: the count is still on the device (a tiny pass launches this kernel without having looked at it: ncand is then the capacity of the list);
    // counters_out: page-locked host memory that receives the pass's counter block (ncand_dev - 1 .. + 11) -- with mlen / mstart in host memory
    // too, such a pass needs no copy at all
    if (counters_out && blockIdx.x == 0 && threadIdx.x < 12) counters_out[threadIdx.x] = ncand_dev[(int)threadIdx.x - 1];
    if (ncand_dev) ncand = min(ncand, *ncand_dev);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint64_t spanmask = (sh.span >= 64) ? ~0ULL : ((1ULL << sh.span) - 1ULL);
    const int N = tab.nseq;
    ExtComp *comp = s_comp[wv];
#ifdef MAUVE_EXT_STATS
    const unsigned long long st_t0 = __builtin_amdgcn_s_memrealtime(); unsigned long long st_rounds = 0, st_setup = 0, st_left = 0, st_right = 0;
#endif
    for (uint32_t ci = wave_global; ci < ncand; ci += nwaves) {
#ifdef MAUVE_EXT_STATS
        unsigned long long st_r = 0; const unsigned long long st_c0 = __builtin_amdgcn_s_memrealtime(); unsigned long long st_c1 = st_c0, st_c2 = st_c0;
#define EXT_STAT_ROUND st_r++
#else
#define EXT_STAT_ROUND
#endif
        const uint32_t ap = cand[ci];
        const uint32_t mask = tmask[ap];
        const int anchor = __ffs(mask) - 1;
        const int nc = __popc(mask);
        const uint32_t segid = SEG ? seg_of(seg + (size_t)anchor * (nseg + 1), nseg, ap - tab.gpos_off[anchor]) : 0u;
        // component table: lane g fills the entry of genome g (entries in ascending genome order, anchor first)
        const uint32_t sa = tpos[(size_t)ap * N + anchor] >> 31;
        if (lane < N && (mask >> lane & 1)) {
            const int g = lane;
            const uint32_t vg = tpos[(size_t)ap * N + g];
            ExtComp C;
            C.G = packed + tab.word_off[g];
            C.VM = vmask ? vmask + tab.mask_off[g] : nullptr;
            C.CM = cmask ? cmask + tab.mask_off[g] : nullptr;
            C.pos = (int64_t)((vg & 0x7fffffffu) - tab.gpos_off[g]);
            C.rev = (vg >> 31) ^ sa; C.g = (uint32_t)g; C.pad = 0;
            C.lo = 0; C.hi = (int64_t)tab.nwin[g] - 1;
            C.maxw = (uint32_t)(((uint64_t)tab.nwin[g] + (uint32_t)sh.span - 1 + 31) / 32 + 2);       // mauve_packed_words - 1
            if (SEG) { const uint32_t *sg = seg + (size_t)g * (nseg + 1) + segid; C.lo = sg[0]; C.hi = (int64_t)sg[1] - sh.span; }
            comp[__popc(mask & ((1u << g) - 1u))] = C;
        }
        __threadfence_block();          // the table is read by every lane of this wave
#ifdef MAUVE_EXT_STATS
        st_c1 = __builtin_amdgcn_s_memrealtime(); st_setup += st_c1 - st_c0;
#endif
        int64_t klo = 0, khi = 0;
        bool leftmost = true;
        if (extend) {
            // offsets every component's window stays inside its genome / gap segment for
            int64_t kmin = INT64_MIN, kmax = INT64_MAX;
            for (int c = 0; c < nc; c++) {
                const ExtComp &C = comp[c];
                const int64_t a = C.rev ? C.pos - C.hi : C.lo - C.pos, b = C.rev ? C.pos - C.lo : C.hi - C.pos;
                kmin = max(kmin, a); kmax = min(kmax, b);
            }
            const bool masks = vmask || cmask;
            // the first round of either walk: offsets -64 .. -1 and 1 .. 64, fetched together
            uint64_t M2[2][4];
            { const int64_t jj[2] = {-64, 1}; ext_mismatch<2>(comp, nc, sh.span, jj, lane, M2); }
            // ---- left walk: offsets cur-1 .. cur-64 per round ----
            int64_t cur = 0;
            for (bool done = false; !done;) {
                const int64_t k = cur - 1 - lane;
                const int64_t hq = (int64_t)ap + k;
                const uint32_t hm = (hq >= 0) ? tmask[hq] : 0u;       // issued beside the genome words
                EXT_STAT_ROUND;
                uint64_t M1[1][4];
                if (cur != 0) { const int64_t jj[1] = {cur - 64}; ext_mismatch<1>(comp, nc, sh.span, jj, lane, M1); }
                else { for (int q = 0; q < 4; q++) M1[0][q] = M2[0][q]; }
                bool a = k >= kmin && k <= kmax && ext_window_clean(M1[0], 63 - lane, sh);
                if (masks) a = a && !ext_blocked(comp, nc, sh.span, k);
                const bool hh = a && hm == mask;
                const uint64_t A = __ballot(a), H = __ballot(hh);
                const int p = walk_round(A, sh.span, done);                          // offsets consumed in this round
                if (H & (p >= 64 ? ~0ULL : ((1ULL << p) - 1ULL))) { leftmost = false; done = true; }      // a same-mask hit among the visited offsets
                cur -= p;
                if (p == 0) done = true;                   // (p == 0 only happens with done set; keeps cur != 0 a valid "not the first round" test)
            }
            if (!leftmost) {
#ifdef MAUVE_EXT_STATS
                st_rounds += st_r; st_left += __builtin_amdgcn_s_memrealtime() - st_c1;
#endif
                if (lane == 0) mlen[ci] = 0; __threadfence_block(); continue;
            }
            klo = cur;
#ifdef MAUVE_EXT_STATS
            st_c2 = __builtin_amdgcn_s_memrealtime(); st_left += st_c2 - st_c1;
#endif
            // ---- right walk ----
            cur = 0;
            for (bool done = false; !done;) {
                const int64_t k = cur + 1 + lane;
                EXT_STAT_ROUND;
                uint64_t M1[1][4];
                if (cur != 0) { const int64_t jj[1] = {cur + 1}; ext_mismatch<1>(comp, nc, sh.span, jj, lane, M1); }
                else { for (int q = 0; q < 4; q++) M1[0][q] = M2[1][q]; }
                bool a = k >= kmin && k <= kmax && ext_window_clean(M1[0], lane, sh);
                if (masks) a = a && !ext_blocked(comp, nc, sh.span, k);
                const uint64_t A = __ballot(a);
                const int p = walk_round(A, sh.span, done);
                cur += p;
            }
            khi = cur;
        }
        if (lane == 0) mlen[ci] = (int32_t)(khi - klo) + sh.span;
        if (lane < N) {
            int32_t st = 0;
            if (mask >> lane & 1) {
                const ExtComp &C = comp[__popc(mask & ((1u << lane) - 1u))];
                st = C.rev ? (int32_t)(-(C.pos - khi + 1)) : (int32_t)(C.pos + klo + 1);
            }
            mstart[(size_t)ci * N + lane] = st;
        }
        __threadfence_block();          // the next candidate overwrites the table
#ifdef MAUVE_EXT_STATS
        st_rounds += st_r; st_right += __builtin_amdgcn_s_memrealtime() - st_c2;
#endif
    }
#ifdef MAUVE_EXT_STATS
    if (lane == 0) {
        const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - st_t0;
        atomicAdd(&g_ext_stats[0], st_rounds); atomicAdd(&g_ext_stats[2], dt); atomicMax(&g_ext_stats[3], dt); atomicAdd(&g_ext_stats[4], 1ULL); atomicMax(&g_ext_stats[5], st_rounds);
        atomicAdd(&g_ext_stats[1], st_setup); atomicAdd(&g_ext_stats[6], st_left); atomicAdd(&g_ext_stats[7], st_right);
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// canonical order on the device (large candidate sets): key = first component << pos_bits | |its start| per candidate
// (pos_bits = bits of the longest genome: fewer radix passes than a fixed 32)
// (dropped candidates get first component = nseq and sort behind everything), the radix sort above on
// (key, candidate index), then a gather of the surviving records as int64 in sorted order.
// ------------------------------------------------------------------------------------------------
// guide tree (progressive.cpp): all it needs of the pairwise matches is the sum of their lengths per genome pair -- no
// canonical order, no copy of the (hundreds of thousands of) records.  Block-level sums in LDS, then one atomic per pair.
__global__ void __launch_bounds__(256) pair_length_sums(const int32_t *__restrict__ mlen, const int32_t *__restrict__ mstart, uint32_t ncand,
                                                        int nseq, unsigned long long *__restrict__ sums)
{
    __shared__ unsigned long long s[MAUVE_MAX_SEQ * MAUVE_MAX_SEQ];
    for (int i = threadIdx.x; i < nseq * nseq; i += 256) s[i] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < ncand; i += gridDim.x * 256u) {
        const int32_t len = mlen[i];
        if (len == 0) continue;
        int a = -1, b = -1;
        for (int g = 0; g < nseq; g++) if (mstart[(size_t)i * nseq + g]) { if (a < 0) a = g; else if (b < 0) b = g; }
        if (b >= 0) atomicAdd(&s[a * nseq + b], (unsigned long long)len);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nseq * nseq; i += 256) if (s[i]) atomicAdd(&sums[i], s[i]);
}

// ---- pairwise breakpoint estimate (progressive.cpp scales node weights by it) ----
// Of every genome pair's matches (length >= min_len): order by position in the lower genome, rank by position in the higher one,
// and count the adjacencies that are not conserved.  Small kernels around three stable radix sorts of (pair, position) keys: by the
// higher genome first (position, strand), so that the order along the lower genome breaks its ties that way, then the ranks.
__device__ __forceinline__ bool bp_pair_of(const int32_t *__restrict__ st, int nseq, int *a, int *b)
{
    int x = -1, y = -1;
    for (int g = 0; g < nseq; g++) if (st[g]) { if (x < 0) x = g; else if (y < 0) y = g; }
    *a = x; *b = y;
    return y >= 0;
}
// which = 1: key by the higher genome (position, strand bit); 0: by the lower genome.  order == nullptr: record j itself (first sort:
// records that do not count get the pair id nseq * nseq, behind every pair).  vals: the record index (keep_record) or j.
__global__ void __launch_bounds__(256) bp_keys(const int32_t *__restrict__ mlen, const int32_t *__restrict__ mstart, const uint32_t *__restrict__ order, uint32_t n, int nseq,
                                               int pos_bits, int32_t min_len, int which, int keep_record, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                               uint32_t *__restrict__ n_valid)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    bool valid = false;
    if (j < n) {
        const uint32_t i = order ? order[j] : j;
        uint64_t key = (uint64_t)(nseq * nseq) << (pos_bits + 1);
        const int32_t len = mlen[i];
        int a, b;
        if (len != 0 && len >= min_len && bp_pair_of(mstart + (size_t)i * nseq, nseq, &a, &b)) {
            const int32_t sx = mstart[(size_t)i * nseq + (which ? b : a)];
            key = ((uint64_t)(a * nseq + b) << (pos_bits + 1)) | ((uint64_t)(sx < 0 ? -sx : sx) << 1) | (uint64_t)(sx < 0);
            valid = true;
        }
        keys[j] = key; vals[j] = keep_record ? i : j;
    }
    if (n_valid) { const uint64_t bal = __ballot(valid); if ((threadIdx.x & 63) == 0 && bal) atomicAdd(n_valid, (uint32_t)__popcll(bal)); }
}
__global__ void __launch_bounds__(256) bp_rank(const uint32_t *__restrict__ order_b, uint32_t nv, uint32_t *__restrict__ rank)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t < nv) rank[order_b[t]] = t;
}
__global__ void __launch_bounds__(256) bp_count(const uint64_t *__restrict__ keys_a, const uint32_t *__restrict__ order_a, const uint32_t *__restrict__ rank,
                                                const int32_t *__restrict__ mstart, uint32_t nv, int nseq, int pos_bits, unsigned long long *__restrict__ out)
{
    __shared__ uint32_t s[MAUVE_MAX_SEQ * MAUVE_MAX_SEQ];
    for (int i = threadIdx.x; i < nseq * nseq; i += 256) s[i] = 0;
    __syncthreads();
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j + 1 < nv; j += gridDim.x * 256u) {
        const uint32_t pair = (uint32_t)(keys_a[j] >> (pos_bits + 1));
        if ((uint32_t)(keys_a[j + 1] >> (pos_bits + 1)) != pair) continue;
        const uint32_t b = pair % (uint32_t)nseq;
        const int32_t s0 = mstart[(size_t)order_a[j] * nseq + b], s1 = mstart[(size_t)order_a[j + 1] * nseq + b];
        const uint32_t r0 = rank[j], r1 = rank[j + 1];
        const bool conserved = (s0 > 0 && s1 > 0 && r1 == r0 + 1) || (s0 < 0 && s1 < 0 && r1 + 1 == r0);
        if (!conserved) atomicAdd(&s[pair], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nseq * nseq; i += 256) if (s[i]) atomicAdd(&out[i], (unsigned long long)s[i]);
}

__global__ void __launch_bounds__(256) canon_keys(const int32_t *__restrict__ mlen, const int32_t *__restrict__ mstart, uint32_t ncand,
                                                  int nseq, int pos_bits, int inval, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                  uint32_t *__restrict__ n_valid)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool valid = false;
    if (i < ncand) {
        uint64_t key = (uint64_t)inval << pos_bits;                 // dropped candidates: behind every match
        if (mlen[i] != 0) {
            const int32_t *s = mstart + (size_t)i * nseq;
            int f = 0; while (f < nseq && s[f] == 0) f++;
            const uint32_t a = f < nseq ? (uint32_t)(s[f] < 0 ? -s[f] : s[f]) : 0u;
            key = ((uint64_t)f << pos_bits) | a;
            valid = true;
        }
        keys[i] = key; vals[i] = i;
    }
    const uint64_t b = __ballot(valid);
    if (b && (threadIdx.x & 63) == (uint32_t)(__ffsll((unsigned long long)b) - 1)) atomicAdd(n_valid, (uint32_t)__popcll(b));
}

// The number of surviving records is still on the device (*n_valid): the launch covers all candidates, the output is
// out[0 .. nm) lengths followed by nm * nseq starts.  Two neighbours with the same key (first component, start) are a
// tie the key alone does not order: *ties is raised and the host finishes the order (rare).
__global__ void __launch_bounds__(256) canon_gather(const int32_t *__restrict__ mlen, const int32_t *__restrict__ mstart,
                                                    const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                    const uint32_t *__restrict__ n_valid, int nseq, int64_t *__restrict__ out,
                                                    uint32_t *__restrict__ ties)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nm = *n_valid;
    if (r >= nm) return;
    int64_t *out_len = out, *out_start = out + nm;
    if (r > 0 && keys[r] == keys[r - 1]) atomicOr(ties, 1u);
    const uint32_t src = vals[r];
    out_len[r] = mlen[src];
    for (int g = 0; g < nseq; g++) out_start[(size_t)r * nseq + g] = mstart[(size_t)src * nseq + g];
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// candidates from which the canonical order is made on the device (below: the host sorts; MAUVE_CANON_DEVICE_MIN: tests force the device path)
static uint32_t canon_device_min()
{
    static const uint32_t v = getenv("MAUVE_CANON_DEVICE_MIN") ? (uint32_t)atol(getenv("MAUVE_CANON_DEVICE_MIN")) : 16384u;
    return v;
}
bool make_seed_shape(uint64_t pattern, SeedShape *sh)
{
    memset(sh, 0, sizeof *sh);
    int span = mauve_seed_length(pattern), w = mauve_seed_weight(pattern);
    if (span < 1 || span > MAUVE_MAX_SEED_SPAN || w < 1 || w > 31) return false;
    // palindromic, first and last set
    for (int t = 0; t < span; t++)
        if (((pattern >> t) & 1) != ((pattern >> (span - 1 - t)) & 1)) return false;
    if (!(pattern & 1)) return false;
    sh->span = span; sh->weight = w; sh->keymask = (1ULL << (2 * w)) - 1ULL;
    int j = 0, nr = 0;
    for (int t = 0; t < span;) {
        if (!((pattern >> (span - 1 - t)) & 1)) { t++; continue; }
        int t0 = t;
        while (t < span && ((pattern >> (span - 1 - t)) & 1)) t++;
        int len = t - t0;
        if (nr >= 32) return false;
        sh->run_src[nr] = (uint8_t)(2 * t0); sh->run_bits[nr] = (uint8_t)(2 * len); sh->run_dst[nr] = (uint8_t)(2 * j);
        nr++; j += len;
    }
    sh->nruns = nr;
    for (int t = 0; t < span; t++) {
        if (!((pattern >> (span - 1 - t)) & 1)) continue;
        if (2 * t < 64) sh->care_lo |= 3ULL << (2 * t); else sh->care_hi |= 3ULL << (2 * t - 64);
    }
    return true;
}

static int build_tab(mauve_ctx *ctx, const GenomeSet &gs, int span, GenomeTab *tab, int64_t *total_windows)
{
    memset(tab, 0, sizeof *tab);
    tab->nseq = gs.nseq;
    int64_t tot = 0;
    for (int g = 0; g < gs.nseq; g++) {
        int64_t nw = gs.lens[g] - span + 1; if (nw < 0) nw = 0;
        tab->gpos_off[g] = (uint32_t)tot; tab->nwin[g] = (uint32_t)nw; tab->word_off[g] = gs.word_off[g];
        tab->mask_off[g] = (gs.vmask || gs.cmask) ? gs.mask_off[g] : 0;
        tot += nw;
        if (tot >= (1LL << 31)) { ctx->err = "total genome length exceeds 2^31 windows"; return MAUVE_ERR_LIMIT; }
    }
    tab->gpos_off[gs.nseq] = (uint32_t)tot;
    for (int g = gs.nseq + 1; g <= MAUVE_MAX_SEQ; g++) tab->gpos_off[g] = 0xffffffffu;      // see genome_of
    *total_windows = tot;
    return MAUVE_OK;
}

template <typename KeyT>
static int sort_pairs(mauve_ctx *ctx, uint32_t n, int key_bits, KeyT **keys_io, uint32_t **vals_io, KeyT *keys_alt,
                      uint32_t *vals_alt, bool have_hist0, int timer_id = -1, int shift_lo = 0)
{
    // LSD passes over bits [shift_lo, key_bits); have_hist0: the tile histograms of the first pass are already in ctx->hist
    // timer_id >= 0: all launches are booked under that id (the small canonical-order sort must not dilute the
    // per-kernel figures of the main sort)
    const int k_hist = timer_id >= 0 ? timer_id : MAUVE_K_SORT_HIST, k_scan = timer_id >= 0 ? timer_id : MAUVE_K_SORT_SCAN,
              k_scat = timer_id >= 0 ? timer_id : MAUVE_K_SORT_SCATTER;
    uint32_t nblk = (n + RS_TILE - 1) / RS_TILE;
    HIPCHK(ctx, ctx->hist.ensure((size_t)nblk * 256 * sizeof(uint32_t)));
    HIPCHK(ctx, ctx->totals.ensure(256 * sizeof(uint32_t)));
    KeyT *kin = *keys_io, *kout = keys_alt; uint32_t *vin = *vals_io, *vout = vals_alt;
    for (int shift = shift_lo; shift < key_bits; shift += 8) {
        if (!(shift == shift_lo && have_hist0)) { KernelTimer t(ctx, k_hist, n);
          hipLaunchKernelGGL(rs_hist<KeyT>, dim3(nblk), dim3(RS_THREADS), 0, ctx->stream, kin, n, shift,
                             ctx->hist.as<uint32_t>(), nblk); }
        static const bool no_raw = getenv("MAUVE_SORT_ROWSCAN") != nullptr;        // A/B switch
        if (nblk <= 64 && !no_raw) {      // small sort: no row scan launch, the scatter reads the raw tile histograms
            KernelTimer t(ctx, k_scat, n);
            hipLaunchKernelGGL((rs_scatter<KeyT, true>), dim3(nblk), dim3(RS_THREADS), 0, ctx->stream, kin, vin, kout, vout, n,
                               shift, ctx->hist.as<uint32_t>(), ctx->totals.as<uint32_t>(), nblk);
            std::swap(kin, kout); std::swap(vin, vout);
            continue;
        }
        { KernelTimer t(ctx, k_scan, n);
          hipLaunchKernelGGL(rs_rowscan, dim3(256), dim3(256), 0, ctx->stream, ctx->hist.as<uint32_t>(), nblk,
                             ctx->totals.as<uint32_t>()); }
        { KernelTimer t(ctx, k_scat, n);
          hipLaunchKernelGGL(rs_scatter<KeyT>, dim3(nblk), dim3(RS_THREADS), 0, ctx->stream, kin, vin, kout, vout, n,
                             shift, ctx->hist.as<uint32_t>(), ctx->totals.as<uint32_t>(), nblk); }
        std::swap(kin, kout); std::swap(vin, vout);
    }
    HIPCHK(ctx, hipGetLastError());
    *keys_io = kin; *vals_io = vin;
    return MAUVE_OK;
}

// One seed pass over a genome set.  SEG: the set is segmented (recursive anchoring); seg = device array
// [nseq][nseg+1] of segment starts.  Results land in ctx->match_len / match_start (canonical order).
template <typename KeyT, bool SEG>
static int seedpass_impl(mauve_ctx *ctx, const GenomeSet &gs, const SeedShape &sh, const GenomeTab &tab, int64_t total,
                         int mode, uint64_t mask, int extend, int only_seq, const uint32_t *seg, uint32_t nseg,
                         int64_t *n_matches, std::vector<uint64_t> *out_keys, std::vector<uint32_t> *out_vals)
{
    const uint32_t n = (uint32_t)total;
    double trace_t0 = now_ms();
    HIPCHK(ctx, ctx->keysA.ensure((size_t)n * sizeof(KeyT)));
    HIPCHK(ctx, ctx->keysB.ensure((size_t)n * sizeof(KeyT)));
    HIPCHK(ctx, ctx->valsA.ensure((size_t)n * 4));
    HIPCHK(ctx, ctx->valsB.ensure((size_t)n * 4));
    HIPCHK(ctx, ctx->counters.ensure(64));
    KeyT *keys = ctx->keysA.as<KeyT>(); uint32_t *vals = ctx->valsA.as<uint32_t>();
    const uint64_t *packed = gs.buf->as<uint64_t>();
    const uint64_t *vmask = gs.vmask ? gs.vmask->as<uint64_t>() : nullptr;
    const uint64_t *cmask = gs.cmask ? gs.cmask->as<uint64_t>() : nullptr;
    const bool masked = vmask || cmask;              // some windows are unusable: placed or ambiguous bases, contig joins

    uint32_t sorted_n = 0;
    bool have_hist0 = false;
    bool compacted = false;
    // Which join: the LDS hash join over a partial sort (join_hash) for the one-pass finders; the full sort and the
    // serial join for PairwiseMatchFinder (its run list needs sorted order), for the sorted-mer-list export and for
    // slices join_hash hands back.  MAUVE_OLD_JOIN forces the second (A/B switch).
    static const bool force_old_join = getenv("MAUVE_OLD_JOIN") != nullptr;
    const HostHits *hh = ctx->host_hits;                      // mauve_extend_hits: the hits come from the host, no sort, no join
    const bool hash_path = !hh && mode != MAUVE_MODE_PAIRWISE && !out_keys && !ctx->enum_req && only_seq < 0 && !force_old_join && !(masked && SEG);
    // segmented keys: segment id above the mer; ids 0 .. nseg-1, the all-ones id is left to the invalid (all-ones) key
    int segbits = 0;
    if (SEG) while (segbits < 32 && (1ull << segbits) <= (uint64_t)nseg) segbits++;
    if (SEG && 2 * sh.weight + segbits > 64) { ctx->err = "recursive anchoring: segment id and mer do not fit 64 bits"; return MAUVE_ERR_LIMIT; }
    const int full_bits = SEG ? 2 * sh.weight + segbits : ((masked && (SEG || only_seq >= 0)) ? (int)sizeof(KeyT) * 8 : 2 * sh.weight);
    // globally sorted bits: 8-bit passes until a bucket averages <= 512 entries; a segmented list sorts at least the
    // segment id, so that the invalid windows form the last bucket on their own
    auto low_bits = [&](uint32_t entries) {
        if (!hash_path) return 0;
        int G = 0;
        while (G < full_bits && ((uint64_t)entries >> G) > 512) G += 8;
        if (SEG) G = std::max(G, (segbits + 7) / 8 * 8);
        return full_bits - std::min(G, full_bits);
    };
    // a pass of a few thousand windows: no key arrays at all (tiny_join).  MAUVE_NO_TINY: A/B switch
    static const bool no_tiny = getenv("MAUVE_NO_TINY") != nullptr;
    const bool tiny = hash_path && only_seq < 0 && !no_tiny && n <= (uint32_t)TJ_MAX && tab.nseq <= 16 && 2 * sh.weight + (SEG ? segbits : 0) <= 32;
    if (hh) sorted_n = 1;
    else if (tiny) sorted_n = n;
    else if (only_seq < 0 && masked && !SEG) {
        // masked pass: only the valid windows go into the sort (see valid_count / seed_extract_compact)
        const uint32_t nblk = (n + 4095) / 4096;
        HIPCHK(ctx, ctx->hist.ensure((size_t)nblk * 256 * sizeof(uint32_t)));     // the tile counts live in the histogram buffer
        uint32_t *tile_cnt = ctx->hist.as<uint32_t>();
        {
            KernelTimer t(ctx, MAUVE_K_EXTRACT, n);
            hipLaunchKernelGGL(valid_count, dim3(nblk), dim3(256), 0, ctx->stream, tab, sh.span, n, vmask, tile_cnt, cmask);
            hipLaunchKernelGGL(tile_scan, dim3(1), dim3(256), 0, ctx->stream, tile_cnt, nblk, ctx->counters.as<uint32_t>());
            if (sh.span <= 32 && sh.weight <= 15)
                hipLaunchKernelGGL((seed_extract_compact<KeyT, true>), dim3(nblk), dim3(256), 0, ctx->stream, packed, tab, sh, keys, vals, n,
                                   vmask, tile_cnt, cmask);
            else
                hipLaunchKernelGGL((seed_extract_compact<KeyT, false>), dim3(nblk), dim3(256), 0, ctx->stream, packed, tab, sh, keys, vals, n,
                                   vmask, tile_cnt, cmask);
        }
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, ctx->pin_seed.ensure(64));
        HIPCHK(ctx, hipMemcpyAsync(ctx->pin_seed.p, ctx->counters.p, 16, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        sorted_n = ctx->pin_seed.as<uint32_t>()[0];
        compacted = true;
    } else if (only_seq < 0) {
        const uint32_t nblk = (n + RS_TILE - 1) / RS_TILE;
        HIPCHK(ctx, ctx->hist.ensure((size_t)nblk * 256 * sizeof(uint32_t)));
        KernelTimer t(ctx, MAUVE_K_EXTRACT, n);
        const int hshift = low_bits(n);
        if (sh.span <= 32 && sh.weight <= 15)
            hipLaunchKernelGGL((seed_extract_all<KeyT, SEG, true>), dim3(nblk), dim3(256), 0, ctx->stream, packed, tab, sh, keys,
                               vals, n, seg, nseg, ctx->hist.as<uint32_t>(), nblk, vmask, hshift, cmask);
        else
            hipLaunchKernelGGL((seed_extract_all<KeyT, SEG, false>), dim3(nblk), dim3(256), 0, ctx->stream, packed, tab, sh, keys,
                               vals, n, seg, nseg, ctx->hist.as<uint32_t>(), nblk, vmask, hshift, cmask);
        sorted_n = n; have_hist0 = true;
    } else {
        const int g = only_seq;
        uint32_t nw = tab.nwin[g];
        if (nw) {
            uint32_t blocks = std::min<uint32_t>((nw + 255) / 256, 256 * 16);
            KernelTimer t(ctx, MAUVE_K_EXTRACT, nw);
            hipLaunchKernelGGL((seed_extract<KeyT, SEG>), dim3(blocks), dim3(256), 0, ctx->stream, packed, tab, sh, g, keys,
                               vals, 0u, seg, nseg);
            sorted_n = nw;
        }
    }
    HIPCHK(ctx, hipGetLastError());
    TRACE(ctx, "extract");
    if (sorted_n == 0) { if (n_matches) *n_matches = 0; return MAUVE_OK; }
    const int key_bits = full_bits;
    const int has_invalid = masked && !compacted;
    const uint32_t ns = sorted_n;                   // entries of the sorted list (all windows, or the valid ones)
    const int L = low_bits(ns);                     // the passes order bits [L, key_bits); 0 = full sort
    int rc = (hh || tiny) ? MAUVE_OK : sort_pairs<KeyT>(ctx, sorted_n, key_bits, &keys, &vals, ctx->keysB.as<KeyT>(), ctx->valsB.as<uint32_t>(), have_hist0, -1, L);
    if (rc) return rc;
    TRACE(ctx, "sort");

    if (ctx->enum_req) {   // SeedMatchEnumerator: runs -> matches on the device, only the CSR result goes to the host
        EnumRequest &q = *ctx->enum_req;
        using namespace devscan;
        const uint32_t nb = (sorted_n + TILE - 1) / TILE, blocks = (sorted_n + 255) / 256;
        HIPCHK(ctx, ctx->run_sum.ensure((size_t)sorted_n * 4 + 64 + ((size_t)sorted_n + 2) * 8 * 3 + (size_t)nb * 16 + 256));
        uint32_t *emit = ctx->run_sum.as<uint32_t>();
        int64_t *soff = reinterpret_cast<int64_t *>(ctx->run_sum.as<char>() + (((size_t)sorted_n * 4 + 63) & ~(size_t)63));
        int64_t *d_mult = soff + sorted_n + 2, *d_off = d_mult + sorted_n + 2, *bsum = d_off + sorted_n + 2, *tot = bsum + nb + 2;
        uint32_t *bcnt = reinterpret_cast<uint32_t *>(tot + 4);
        hipLaunchKernelGGL((enum_runs<KeyT>), dim3(blocks), dim3(256), 0, ctx->stream, keys, vals, sorted_n, q.min_multi, q.max_multi, q.direct_only, emit);
        hipLaunchKernelGGL((vscan_partial<int64_t, EmitVal>), dim3(nb), dim3(256), 0, ctx->stream, EmitVal{emit}, sorted_n, bsum);
        hipLaunchKernelGGL((vscan_write<int64_t, EmitVal>), dim3(nb), dim3(256), 0, ctx->stream, EmitVal{emit}, sorted_n, bsum, soff, tot + 1);
        const EnumRuns er{emit, soff, sorted_n, d_mult, d_off, tot};
        hipLaunchKernelGGL((cmp_count<EnumRuns>), dim3(nb), dim3(256), 0, ctx->stream, er, bcnt);
        hipLaunchKernelGGL((cmp_write<EnumRuns>), dim3(nb), dim3(256), 0, ctx->stream, er, bcnt);
        HIPCHK(ctx, hipGetLastError());
        int64_t ht[2] = {0, 0};
        HIPCHK(ctx, hipMemcpyAsync(ht, tot, 16, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        q.n = ht[0]; q.ns = ht[1];
        if (q.starts) {
            HIPCHK(ctx, ctx->sorted_rec.ensure(((size_t)q.ns + 1) * 8));
            hipLaunchKernelGGL((enum_write<KeyT>), dim3(blocks), dim3(256), 0, ctx->stream, keys, vals, sorted_n, q.direct_only, emit, soff,
                               tab.gpos_off[only_seq], ctx->sorted_rec.as<int64_t>());
            HIPCHK(ctx, hipGetLastError());
            HIPCHK(ctx, hipMemcpyAsync(q.mult, d_mult, (size_t)q.n * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipMemcpyAsync(q.start_off, d_off, ((size_t)q.n + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipMemcpyAsync(q.starts, ctx->sorted_rec.p, (size_t)q.ns * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        }
        return MAUVE_OK;
    }
    if (out_keys) {   // sorted-mer-list export: hand the sorted pairs to the host
        std::vector<KeyT> hk(sorted_n);
        out_vals->resize(sorted_n);
        HIPCHK(ctx, hipMemcpyAsync(hk.data(), keys, (size_t)sorted_n * sizeof(KeyT), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(out_vals->data(), vals, (size_t)sorted_n * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        out_keys->resize(sorted_n);
        for (uint32_t i = 0; i < sorted_n; i++) (*out_keys)[i] = (uint64_t)hk[i];
        return MAUVE_OK;
    }

    // ---- join + extension, once per finder pass (one pass, or one per genome pair for PairwiseMatchFinder) ----
    const int N = tab.nseq;
    const uint32_t P = n;
    HIPCHK(ctx, ctx->posmask.ensure((size_t)P * 4));             // tmask
    HIPCHK(ctx, ctx->hit_pos.ensure((size_t)P * 4 * N));         // tpos [P][N]
    // Capacities of the compacted lists are handed to the kernels that fill them (a store past the end is skipped, the counter still counts)
    // and checked against the counts that come back.  MAUVE_LIST_CAP: test knob, shrinks them so that the check can be seen to fire.
    static const uint32_t list_cap_env = getenv("MAUVE_LIST_CAP") ? (uint32_t)strtoul(getenv("MAUVE_LIST_CAP"), nullptr, 10) : 0u;
    uint32_t cand_cap = P / 2 + 1;                               // a hit needs >= 2 entries
    HIPCHK(ctx, ctx->cand.ensure((size_t)cand_cap * 4));
    if (list_cap_env) cand_cap = std::min(cand_cap, list_cap_env);
    auto cand_overflow = [&](uint32_t nc) { if (nc <= cand_cap) return false; ctx->err = "seed pass: " + std::to_string(nc) + " candidates for a list of " + std::to_string(cand_cap); return true; };
    uint32_t *tmask = ctx->posmask.as<uint32_t>(), *tpos = ctx->hit_pos.as<uint32_t>();
    struct FinderPass { uint32_t consider, want; int rule; };
    std::vector<FinderPass> passes;
    if (mode == MAUVE_MODE_PAIRWISE) {
        // (guide tree over several contexts, mauve_set_shard: this rank's share of the pairs, dealt round robin; the sums are exchanged)
        int pi = 0;
        for (int i = 0; i < N; i++) for (int j = i + 1; j < N; j++, pi++)
            if (!ctx->pair_sums_only || !ctx->shard_on || pi % ctx->shard_world == ctx->shard_rank)
                passes.push_back({(1u << i) | (1u << j), (1u << i) | (1u << j), MAUVE_MODE_MEM});
    } else passes.push_back({0xffffffffu, (uint32_t)mask, mode});
    ctx->n_matches = 0; ctx->match_len.clear(); ctx->match_start.clear(); ctx->matches_pending = false;
    if (n_matches) *n_matches = 0;
    // one record slot per candidate of every pass; length 0 = not a leftmost hit (host scratch kept across calls)
    std::vector<int32_t> &hl = ctx->sdh.hl, &hs = ctx->sdh.hs; hl.clear(); hs.clear();
    uint32_t cand_total = 0;
    bool records_on_host = false;                 // hl / hs hold the candidates' records already (the one-round-trip form of a tiny pass)
    // pairwise mode: the runs that can matter to any pair, listed once (see run_summary)
    uint32_t nruns = 0;
    const bool use_summary = mode == MAUVE_MODE_PAIRWISE && !SEG && (passes.size() > 1 || (ctx->pair_sums_only && ctx->shard_on && !passes.empty()));
    uint32_t *rstart = nullptr, *rlen = nullptr, *runiq = nullptr;
    if (use_summary) {
        const size_t cap = (size_t)ns / 2 + 1;
        const uint32_t run_cap = list_cap_env ? std::min<uint32_t>((uint32_t)cap, list_cap_env) : (uint32_t)cap;
        HIPCHK(ctx, ctx->run_sum.ensure(3 * cap * 4));
        rstart = ctx->run_sum.as<uint32_t>(); rlen = rstart + cap; runiq = rlen + cap;
        HIPCHK(ctx, hipMemsetAsync(ctx->counters.p, 0, 64, ctx->stream));
        { KernelTimer t(ctx, MAUVE_K_JOIN, ns);
          hipLaunchKernelGGL((run_summary<KeyT>), dim3((ns + 1023) / 1024), dim3(256), 0, ctx->stream, keys, vals, ns, tab,
                             has_invalid, rstart, rlen, runiq, ctx->counters.as<uint32_t>() + 2, run_cap); }
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, ctx->pin_seed.ensure(64));
        HIPCHK(ctx, hipMemcpyAsync(ctx->pin_seed.p, ctx->counters.p, 16, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        nruns = ctx->pin_seed.as<uint32_t>()[2];
        if (nruns > run_cap) { ctx->err = "seed pass: " + std::to_string(nruns) + " runs for a list of " + std::to_string(run_cap); return MAUVE_ERR_LIMIT; }
        TRACE(ctx, "run summary");
    }
    // ---- the passes of the pairwise finder in groups: pairs with different lower genomes write disjoint slices of the hit table, so up to
    // N - 1 of them share one join launch (the run list is read once per group instead of once per pair), one run-detection launch
    // (blockIdx.y = pair), one candidate list, one round trip and one extension launch: 28 passes of an 8-genome guide tree are 7 groups.
    // MAUVE_PAIR_SERIAL: A/B switch (one pass per pair, the loop below).
    static const bool pair_serial = getenv("MAUVE_PAIR_SERIAL") != nullptr;
    if (use_summary && nruns && !pair_serial && !hh && !SEG) {
        // one candidate list per group: a pair has at most one hit per window of its lower genome, the lower genomes of a group are
        // different, so a group has at most P candidates (a single pass: P / 2, "a hit needs two entries" -- not enough here)
        HIPCHK(ctx, ctx->cand.ensure(((size_t)P + 1) * 4));
        cand_cap = list_cap_env ? std::min<uint32_t>(P + 1, list_cap_env) : P + 1;
        std::vector<char> used(passes.size(), 0);
        for (size_t left = passes.size(); left;) {
            PairGroup grp; memset(&grp, 0, sizeof grp);
            uint32_t amask = 0, maxslice = 0, slices = 0;
            for (size_t q = 0; q < passes.size(); q++) {
                if (used[q]) continue;
                const int ga = __builtin_ctz(passes[q].consider), gb = 31 - __builtin_clz(passes[q].consider);
                if (amask >> ga & 1u) continue;
                used[q] = 1; left--;
                const uint32_t lo = tab.gpos_off[ga], hi = std::min<uint32_t>(tab.gpos_off[ga + 1], P);
                if (hi <= lo) continue;                            // (a genome shorter than the seed has no window)
                grp.ga[grp.n] = ga; grp.gb[grp.n] = gb; grp.lo[grp.n] = lo; grp.hi[grp.n] = hi; grp.n++;
                amask |= 1u << ga; maxslice = std::max(maxslice, hi - lo); slices += hi - lo;
            }
            if (!grp.n) continue;
            for (int y = 0; y < grp.n; y++) HIPCHK(ctx, hipMemsetAsync(ctx->posmask.as<uint32_t>() + grp.lo[y], 0, (size_t)(grp.hi[y] - grp.lo[y]) * 4, ctx->stream));
            HIPCHK(ctx, hipMemsetAsync(ctx->counters.p, 0, 64, ctx->stream));
            { KernelTimer t(ctx, MAUVE_K_JOIN, nruns);
              hipLaunchKernelGGL(join_pair_group, dim3((nruns + 255) / 256), dim3(256), 0, ctx->stream, vals, tab, rstart, rlen, runiq, nruns, grp, tmask, tpos); }
            HIPCHK(ctx, hipGetLastError());
            TRACE(ctx, "join");
            { KernelTimer t(ctx, MAUVE_K_RUNS, slices);
              hipLaunchKernelGGL(mum_runs_group, dim3((maxslice + RUNS_TILE - 1) / RUNS_TILE, (uint32_t)grp.n), dim3(256), 0, ctx->stream, tab, sh.span, tmask, tpos, grp,
                                 extend ? 0 : 1, ctx->cand.as<uint32_t>(), ctx->counters.as<uint32_t>(), cand_cap); }
            HIPCHK(ctx, hipGetLastError());
            if (ctx->shadow) { std::function<void()> fsh; fsh.swap(ctx->shadow); fsh(); }
            HIPCHK(ctx, ctx->pin_seed.ensure(64));
            HIPCHK(ctx, hipMemcpyAsync(ctx->pin_seed.p, ctx->counters.p, 16, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            const uint32_t nc = ctx->pin_seed.as<uint32_t>()[1];
            if (cand_overflow(nc)) return MAUVE_ERR_LIMIT;
            TRACE(ctx, "runs");
            if (g_trace) fprintf(stderr, "[trace]   %u candidates of %u windows (%d pairs at once)\n", nc, slices, grp.n);
            if (nc == 0) continue;
            HIPCHK(ctx, ctx->mlen.ensure_keep((size_t)(cand_total + nc) * 4 + 4, (size_t)cand_total * 4, ctx->stream));
            HIPCHK(ctx, ctx->mstart.ensure_keep((size_t)(cand_total + nc) * 4 * N + 4, (size_t)cand_total * 4 * N, ctx->stream));
            {
                const uint32_t blocks = std::min<uint32_t>((nc + 3) / 4, 256 * 8);
                KernelTimer t(ctx, MAUVE_K_EXTEND, nc);
                hipLaunchKernelGGL((mum_extend<SEG>), dim3(blocks), dim3(256), 0, ctx->stream, packed, tab, sh, tmask, tpos, P, ctx->cand.as<uint32_t>(), nc, extend,
                                   ctx->mlen.as<int32_t>() + cand_total, ctx->mstart.as<int32_t>() + (size_t)cand_total * N, seg, nseg, vmask, cmask);
                HIPCHK(ctx, hipGetLastError());
            }
            cand_total += nc;
            TRACE(ctx, "extend");
        }
        passes.clear();                                        // done: nothing left for the loop below
    }
    for (const FinderPass &fp : passes) {
        // the hit table is indexed by the anchor's window = a window of the lowest genome of the pass: a pass over one genome pair
        // (the guide tree runs N (N - 1) / 2 of them) clears and scans that genome's slice only
        uint32_t s_lo = 0, s_hi = P;
        if (use_summary && fp.consider) { const int ga = __builtin_ctz(fp.consider); s_lo = tab.gpos_off[ga]; s_hi = std::min<uint32_t>(tab.gpos_off[ga + 1], P); }
        if (s_hi <= s_lo) continue;                              // (a genome shorter than the seed has no window: nothing can be anchored in it)
        const bool tiny_clears = tiny && !hh && !use_summary;           // tiny_join clears for itself
        if (!tiny_clears) {
            HIPCHK(ctx, hipMemsetAsync(ctx->posmask.as<uint32_t>() + s_lo, 0, (size_t)(s_hi - s_lo) * 4, ctx->stream));
            HIPCHK(ctx, hipMemsetAsync(ctx->counters.p, 0, 64, ctx->stream));
        }
        if (hh) {
            HIPCHK(ctx, ctx->run_sum.ensure((size_t)hh->n * (N + 1) * 4 + 64));
            HIPCHK(ctx, hipMemcpyAsync(ctx->run_sum.p, hh->rec, (size_t)hh->n * (N + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
            if (hh->n) hipLaunchKernelGGL(hits_scatter, dim3((hh->n + 255) / 256), dim3(256), 0, ctx->stream, ctx->run_sum.as<uint32_t>(), hh->n, N, P, tmask, tpos);
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));      // the host records must outlive the copy
        } else if (use_summary) {
            KernelTimer t(ctx, MAUVE_K_JOIN, nruns);
            if (nruns)
                hipLaunchKernelGGL(join_pair, dim3((nruns + 255) / 256), dim3(256), 0, ctx->stream, vals, tab, rstart, rlen, runiq, nruns,
                                   __builtin_ctz(fp.consider), 31 - __builtin_clz(fp.consider), tmask, tpos);
        } else if (tiny) {
            static const bool tj_attr = []() { return hipFuncSetAttribute(reinterpret_cast<const void *>(tiny_join), hipFuncAttributeMaxDynamicSharedMemorySize, TJ_SLOTS * 8) == hipSuccess; }();
            if (!tj_attr) { ctx->err = "tiny_join: cannot reserve its LDS"; return MAUVE_ERR_HIP; }
            KernelTimer t(ctx, MAUVE_K_JOIN, P);
            hipLaunchKernelGGL(tiny_join, dim3(1), dim3(1024), TJ_SLOTS * 8, ctx->stream, packed, tab, sh, P, vmask, cmask, fp.rule, fp.want, tmask, tpos,
                               ctx->counters.as<uint32_t>() + 9, SEG ? seg : (const uint32_t *)nullptr, nseg, ctx->counters.as<uint32_t>(), s_lo, s_hi);
        } else if (hash_path) {
            const uint32_t nchunk = (ns + HJ_T - 1) / HJ_T;
            HIPCHK(ctx, ctx->join_ovf.ensure((2 + 2 * (size_t)HJ_OVF_CAP) * 4));        // the ranges; their count sits in the counter block (words 8, 9)
            KernelTimer t(ctx, MAUVE_K_JOIN, ns);
            HIPCHK(ctx, ctx->join_bound.ensure(((size_t)nchunk + 2) * 4));
            hipLaunchKernelGGL((join_bounds<KeyT>), dim3((nchunk + 1 + 3) / 4), dim3(256), 0, ctx->stream, keys, ns, L, nchunk,
                               ctx->join_bound.as<uint32_t>());
#define JH_LAUNCH(W) hipLaunchKernelGGL((join_hash<KeyT, W>), dim3(nchunk), dim3(256), 0, ctx->stream, keys, vals, ns, \
                                           ctx->join_bound.as<uint32_t>(), tab, fp.rule, fp.want, tmask, tpos, ctx->counters.as<uint32_t>() + 8, ctx->join_ovf.as<uint32_t>(), P)
            if (N > 16) JH_LAUNCH(true);
            else JH_LAUNCH(false);
#undef JH_LAUNCH
        } else
        { KernelTimer t(ctx, MAUVE_K_JOIN, ns);
          hipLaunchKernelGGL((mum_join<KeyT, SEG>), dim3((ns + 255) / 256), dim3(256), 0, ctx->stream, keys, vals, ns, tab, fp.rule,
                             fp.want, fp.consider, tmask, tpos, P, has_invalid); }
        HIPCHK(ctx, hipGetLastError());
        TRACE(ctx, "join");
        // extension phase A: run starts from the table
        { KernelTimer t(ctx, MAUVE_K_RUNS, s_hi - s_lo);
          hipLaunchKernelGGL((mum_runs<SEG>), dim3((s_hi - s_lo + RUNS_TILE - 1) / RUNS_TILE), dim3(256), 0, ctx->stream, tab,
                             sh.span, tmask, tpos, s_hi, extend ? 0 : 1, ctx->cand.as<uint32_t>(), ctx->counters.as<uint32_t>(), seg,
                             nseg, cand_cap, s_lo); }
        HIPCHK(ctx, hipGetLastError());
        if (ctx->shadow) { std::function<void()> f; f.swap(ctx->shadow); f(); }     // the kernels above are still running
        if (tiny && !hh && passes.size() == 1 && cand_total == 0 && cand_cap <= 8192) {
            // A tiny pass (a round of the LCB extension, a small guide-tree node, a small recursion batch) is a handful of 5-10 us kernels: the round
            // trip that fetched the candidate count before the extension kernel could be launched cost as much as the pass.  Its candidate list has
            // at most a few thousand entries, so the extension is launched for the CAPACITY of the list with the count left on the device, and the
            // counters come back together with the records: one synchronisation per pass instead of two.
            // The records and the counters go straight into page-locked host memory (the device writes it in place: a handful of candidates), so the
            // round trip is one synchronisation and no copy kernel.  (Not when the canonical order is to be made on the device -- a test setting for
            // lists this small: then the records stay in device memory and are copied as well.)
            const size_t lbytes = ((size_t)cand_cap * 4 + 63) & ~(size_t)63, sbytes = (size_t)cand_cap * 4 * N;
            HIPCHK(ctx, ctx->pin_seed.ensure(64 + lbytes + sbytes));
            char *pin = ctx->pin_seed.as<char>();
            const bool host_out = cand_cap < canon_device_min();
            if (!host_out) { HIPCHK(ctx, ctx->mlen.ensure((size_t)cand_cap * 4 + 4)); HIPCHK(ctx, ctx->mstart.ensure((size_t)cand_cap * 4 * N + 4)); }
            int32_t *o_len = host_out ? reinterpret_cast<int32_t *>(pin + 64) : ctx->mlen.as<int32_t>();
            int32_t *o_st = host_out ? reinterpret_cast<int32_t *>(pin + 64 + lbytes) : ctx->mstart.as<int32_t>();
            { KernelTimer t(ctx, MAUVE_K_EXTEND, cand_cap);
              hipLaunchKernelGGL((mum_extend<SEG>), dim3(std::min<uint32_t>((cand_cap + 3) / 4, 512)), dim3(256), 0, ctx->stream, packed, tab, sh, tmask, tpos, P,
                                 ctx->cand.as<uint32_t>(), cand_cap, extend, o_len, o_st, seg, nseg, vmask, cmask,
                                 ctx->counters.as<uint32_t>() + 1, reinterpret_cast<uint32_t *>(pin)); }
            HIPCHK(ctx, hipGetLastError());
            if (!host_out) {
                HIPCHK(ctx, hipMemcpyAsync(pin + 64, ctx->mlen.p, (size_t)cand_cap * 4, hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(ctx, hipMemcpyAsync(pin + 64 + lbytes, ctx->mstart.p, sbytes, hipMemcpyDeviceToHost, ctx->stream));
            }
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            const uint32_t nc = reinterpret_cast<const uint32_t *>(pin)[1];
            if (reinterpret_cast<const uint32_t *>(pin)[9]) { ctx->err = "tiny_join: anchor out of range (internal error)"; return MAUVE_ERR_HIP; }
            if (cand_overflow(nc)) return MAUVE_ERR_LIMIT;
            TRACE(ctx, "runs + extend (one round trip)");
            if (g_trace) fprintf(stderr, "[trace]   %u candidates of %u windows\n", nc, P);
            hl.assign(reinterpret_cast<const int32_t *>(pin + 64), reinterpret_cast<const int32_t *>(pin + 64) + nc);
            hs.assign(reinterpret_cast<const int32_t *>(pin + 64 + lbytes), reinterpret_cast<const int32_t *>(pin + 64 + lbytes) + (size_t)nc * N);
            cand_total = nc; records_on_host = true;
            continue;
        }
        HIPCHK(ctx, ctx->pin_seed.ensure(64));
        HIPCHK(ctx, hipMemcpyAsync(ctx->pin_seed.p, ctx->counters.p, 48, hipMemcpyDeviceToHost, ctx->stream));     // run counters + join_hash's overflow count
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        uint32_t nc = ctx->pin_seed.as<uint32_t>()[1];
        const uint32_t novf = hash_path && !tiny ? ctx->pin_seed.as<uint32_t>()[8] : 0u;
        if (hash_path && ctx->pin_seed.as<uint32_t>()[9]) { ctx->err = "join_hash: anchor out of range (internal error)"; return MAUVE_ERR_HIP; }
        if (novf) {
            // ranges join_hash declined (a bucket beyond its LDS table): full sort + serial join of each slice, or of
            // the whole list when there are more of them than the list holds; then the run detection again
            std::vector<uint32_t> rng;
            if (novf > (uint32_t)HJ_OVF_CAP) rng = {0u, ns};
            else {
                rng.resize(2 * (size_t)novf);
                HIPCHK(ctx, hipMemcpy(rng.data(), ctx->join_ovf.as<uint32_t>() + 2, rng.size() * 4, hipMemcpyDeviceToHost));
            }
            KeyT *alt_k = keys == ctx->keysA.as<KeyT>() ? ctx->keysB.as<KeyT>() : ctx->keysA.as<KeyT>();
            uint32_t *alt_v = vals == ctx->valsA.as<uint32_t>() ? ctx->valsB.as<uint32_t>() : ctx->valsA.as<uint32_t>();
            for (size_t q = 0; q + 1 < rng.size(); q += 2) {
                const uint32_t s0 = rng[q], cnt = rng[q + 1] - rng[q];
                KeyT *kp = keys + s0; uint32_t *vp = vals + s0;
                int rc3 = sort_pairs<KeyT>(ctx, cnt, key_bits, &kp, &vp, alt_k + s0, alt_v + s0, false, MAUVE_K_JOIN);
                if (rc3) return rc3;
                KernelTimer t(ctx, MAUVE_K_JOIN, cnt);
                hipLaunchKernelGGL((mum_join<KeyT, SEG>), dim3((cnt + 255) / 256), dim3(256), 0, ctx->stream, kp, vp, cnt, tab, fp.rule,
                                   fp.want, fp.consider, tmask, tpos, P, has_invalid);
            }
            HIPCHK(ctx, hipGetLastError());
            HIPCHK(ctx, hipMemsetAsync(ctx->counters.p, 0, 64, ctx->stream));
            { KernelTimer t(ctx, MAUVE_K_RUNS, P);
              hipLaunchKernelGGL((mum_runs<SEG>), dim3((P + RUNS_TILE - 1) / RUNS_TILE), dim3(256), 0, ctx->stream, tab,
                                 sh.span, tmask, tpos, P, extend ? 0 : 1, ctx->cand.as<uint32_t>(), ctx->counters.as<uint32_t>(), seg,
                                 nseg, cand_cap); }
            HIPCHK(ctx, hipGetLastError());
            HIPCHK(ctx, hipMemcpyAsync(ctx->pin_seed.p, ctx->counters.p, 16, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            nc = ctx->pin_seed.as<uint32_t>()[1];
            if (g_trace) fprintf(stderr, "[trace]   join_hash handed back %u range(s)\n", novf);
        }
        if (cand_overflow(nc)) return MAUVE_ERR_LIMIT;
        TRACE(ctx, "runs");
        if (g_trace) fprintf(stderr, "[trace]   %u candidates of %u windows\n", nc, P);
        if (nc == 0) continue;
        // extension phase B: the records of all passes accumulate on the device
        HIPCHK(ctx, ctx->mlen.ensure_keep((size_t)(cand_total + nc) * 4 + 4, (size_t)cand_total * 4, ctx->stream));
        HIPCHK(ctx, ctx->mstart.ensure_keep((size_t)(cand_total + nc) * 4 * N + 4, (size_t)cand_total * 4 * N, ctx->stream));
        {
            // six waves per SIMD fit (77 registers); two rounds of workgroups even out the candidates' different lengths (measured: 5 .. 8 per compute unit
            // within 5 % of each other, 12 .. 14 another 10 % faster)
            static const int per_cu = getenv("MAUVE_EXT_BLOCKS_PER_CU") ? atoi(getenv("MAUVE_EXT_BLOCKS_PER_CU")) : 12;
            uint32_t blocks = std::min<uint32_t>((nc + 3) / 4, (uint32_t)(ctx->cus * per_cu));
            KernelTimer t(ctx, MAUVE_K_EXTEND, nc);
            hipLaunchKernelGGL((mum_extend<SEG>), dim3(blocks), dim3(256), 0, ctx->stream, packed, tab, sh, tmask, tpos, P,
                               ctx->cand.as<uint32_t>(), nc, extend, ctx->mlen.as<int32_t>() + cand_total,
                               ctx->mstart.as<int32_t>() + (size_t)cand_total * N, seg, nseg, vmask, cmask);
            HIPCHK(ctx, hipGetLastError());
#ifdef MAUVE_EXT_STATS
            if (g_trace) {
                unsigned long long h[8]; hipStreamSynchronize(ctx->stream);
                hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ext_stats), sizeof h); unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_ext_stats), z, sizeof z);
                fprintf(stderr, "[trace]   extend stats: %llu rounds, %llu waves, wave time mean %.1f us max %.1f us (setup %.1f, left walk %.1f, right walk %.1f), most rounds of a wave %llu\n", h[0], h[4],
                        h[4] ? h[2] / (double)h[4] / 100.0 : 0.0, h[3] / 100.0, h[4] ? h[1] / (double)h[4] / 100.0 : 0.0, h[4] ? h[6] / (double)h[4] / 100.0 : 0.0,
                        h[4] ? h[7] / (double)h[4] / 100.0 : 0.0, h[5]);
            }
#endif
        }
        cand_total += nc;
        TRACE(ctx, "extend");
    }
    const uint32_t ncand = cand_total;
    if (ctx->pair_sums_only) {                   // the guide tree's view of the pairwise matches
        ctx->pair_sums.assign((size_t)N * N, 0);
        if (ncand) {
            HIPCHK(ctx, ctx->run_sum.ensure((size_t)N * N * 8 + 64));
            unsigned long long *d = ctx->run_sum.as<unsigned long long>();
            HIPCHK(ctx, hipMemsetAsync(d, 0, (size_t)N * N * 8, ctx->stream));
            hipLaunchKernelGGL(pair_length_sums, dim3(std::min<uint32_t>((ncand + 255) / 256, 1024)), dim3(256), 0, ctx->stream, ctx->mlen.as<int32_t>(),
                               ctx->mstart.as<int32_t>(), ncand, N, d);
            HIPCHK(ctx, hipGetLastError());
            HIPCHK(ctx, ctx->pin_seed.ensure(64 + (size_t)N * N * 8));
            HIPCHK(ctx, hipMemcpyAsync(ctx->pin_seed.as<char>() + 64, d, (size_t)N * N * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            memcpy(ctx->pair_sums.data(), ctx->pin_seed.as<char>() + 64, (size_t)N * N * 8);
        }
        TRACE(ctx, "pair length sums");
        if (ctx->bp_min_len >= 0) {              // broken adjacencies per pair, from the same records
            ctx->pair_bp.assign((size_t)N * N, 0);
            if (ncand >= 2) {
                HIPCHK(ctx, ctx->canon_k1.ensure((size_t)ncand * 8 + 64)); HIPCHK(ctx, ctx->canon_k2.ensure((size_t)ncand * 8 + 64));
                HIPCHK(ctx, ctx->canon_v1.ensure((size_t)ncand * 4 + 64)); HIPCHK(ctx, ctx->canon_v2.ensure((size_t)ncand * 4 + 64));
                HIPCHK(ctx, ctx->bp_work.ensure((size_t)ncand * 8 * 2 + (size_t)ncand * 4 * 3 + (size_t)N * N * 8 + 256));
                uint64_t *ck = ctx->canon_k1.as<uint64_t>(), *ck2 = ctx->canon_k2.as<uint64_t>();
                uint32_t *cv = ctx->canon_v1.as<uint32_t>(), *cv2 = ctx->canon_v2.as<uint32_t>();
                uint64_t *bk = ctx->bp_work.as<uint64_t>(), *bk2 = bk + ncand;
                uint32_t *bv = reinterpret_cast<uint32_t *>(bk2 + ncand), *bv2 = bv + ncand, *rank = bv2 + ncand;
                unsigned long long *dbp = reinterpret_cast<unsigned long long *>(ctx->bp_work.as<char>() + (((size_t)ncand * 28 + 63) & ~(size_t)63));
                HIPCHK(ctx, hipMemsetAsync(ctx->counters.p, 0, 64, ctx->stream));
                HIPCHK(ctx, hipMemsetAsync(dbp, 0, (size_t)N * N * 8, ctx->stream));
                int64_t maxlen = 1; for (int g = 0; g < N; g++) maxlen = std::max<int64_t>(maxlen, gs.lens[(size_t)g]);
                int pos_bits = 1; while (pos_bits < 32 && (1LL << pos_bits) <= maxlen) pos_bits++;
                int pid_bits = 1; while ((1 << pid_bits) <= N * N) pid_bits++;
                const int32_t min_len = (int32_t)std::min<int64_t>(ctx->bp_min_len, INT32_MAX);
                const int kb = pos_bits + 1 + pid_bits;
                // 1. by the higher genome (position, strand): only to break the ties of the next order
                hipLaunchKernelGGL(bp_keys, dim3((ncand + 255) / 256), dim3(256), 0, ctx->stream, ctx->mlen.as<int32_t>(), ctx->mstart.as<int32_t>(), (const uint32_t *)nullptr, ncand, N,
                                   pos_bits, min_len, 1, 1, ck, cv, ctx->counters.as<uint32_t>() + 3);
                HIPCHK(ctx, hipGetLastError());
                int rc2 = sort_pairs<uint64_t>(ctx, ncand, kb, &ck, &cv, ck2, cv2, false, MAUVE_K_CANON);
                if (rc2) return rc2;
                HIPCHK(ctx, hipMemcpyAsync(ctx->pin_seed.as<char>(), ctx->counters.as<uint32_t>() + 3, 4, hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                const uint32_t nv = ctx->pin_seed.as<uint32_t>()[0];
                if (nv >= 2) {
                    // 2. along the lower genome: record indices in that order (stable: ties stay in the order of 1.)
                    hipLaunchKernelGGL(bp_keys, dim3((nv + 255) / 256), dim3(256), 0, ctx->stream, ctx->mlen.as<int32_t>(), ctx->mstart.as<int32_t>(), cv, nv, N, pos_bits, min_len, 0, 1,
                                       bk, bv, (uint32_t *)nullptr);
                    HIPCHK(ctx, hipGetLastError());
                    uint64_t *ak = bk; uint32_t *av = bv;
                    rc2 = sort_pairs<uint64_t>(ctx, nv, kb, &ak, &av, bk2, bv2, false, MAUVE_K_CANON);
                    if (rc2) return rc2;
                    // 3. ranks along the higher genome (ties in the order of 2.); the canonical-sort buffers are free again
                    uint64_t *sk = ctx->canon_k1.as<uint64_t>(); uint32_t *sv = ctx->canon_v1.as<uint32_t>();
                    hipLaunchKernelGGL(bp_keys, dim3((nv + 255) / 256), dim3(256), 0, ctx->stream, ctx->mlen.as<int32_t>(), ctx->mstart.as<int32_t>(), av, nv, N, pos_bits, min_len, 1, 0,
                                       sk, sv, (uint32_t *)nullptr);
                    HIPCHK(ctx, hipGetLastError());
                    rc2 = sort_pairs<uint64_t>(ctx, nv, kb, &sk, &sv, ctx->canon_k2.as<uint64_t>(), ctx->canon_v2.as<uint32_t>(), false, MAUVE_K_CANON);
                    if (rc2) return rc2;
                    ck = ak; cv = av;                        // the order along the lower genome, for the count
                    hipLaunchKernelGGL(bp_rank, dim3((nv + 255) / 256), dim3(256), 0, ctx->stream, sv, nv, rank);
                    hipLaunchKernelGGL(bp_count, dim3(std::min<uint32_t>((nv + 255) / 256, 1024)), dim3(256), 0, ctx->stream, ck, cv, rank, ctx->mstart.as<int32_t>(), nv, N, pos_bits, dbp);
                    HIPCHK(ctx, hipGetLastError());
                    HIPCHK(ctx, ctx->pin_seed.ensure(64 + (size_t)N * N * 8));
                    HIPCHK(ctx, hipMemcpyAsync(ctx->pin_seed.as<char>() + 64, dbp, (size_t)N * N * 8, hipMemcpyDeviceToHost, ctx->stream));
                    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                    memcpy(ctx->pair_bp.data(), ctx->pin_seed.as<char>() + 64, (size_t)N * N * 8);
                }
            }
            TRACE(ctx, "pair breakpoints");
        }
        return MAUVE_OK;
    }
    if (ncand == 0) return MAUVE_OK;
    // ---- canonical order (first component, |start|, mask, starts, length) ----
    const uint32_t dev_sort_min = canon_device_min();
    if (ncand >= dev_sort_min) {
        // large sets: sort on the device, gather, copy out in order.  (Own buffers: over several finder passes the
        // candidates can outnumber the windows, so the sorted-mer buffers are not guaranteed to be big enough.)
        HIPCHK(ctx, ctx->canon_k1.ensure((size_t)ncand * 8 + 64)); HIPCHK(ctx, ctx->canon_k2.ensure((size_t)ncand * 8 + 64));
        HIPCHK(ctx, ctx->canon_v1.ensure((size_t)ncand * 4 + 64)); HIPCHK(ctx, ctx->canon_v2.ensure((size_t)ncand * 4 + 64));
        uint64_t *ck = ctx->canon_k1.as<uint64_t>(), *ck2 = ctx->canon_k2.as<uint64_t>();
        uint32_t *cv = ctx->canon_v1.as<uint32_t>(), *cv2 = ctx->canon_v2.as<uint32_t>();
        HIPCHK(ctx, hipMemsetAsync(ctx->counters.p, 0, 64, ctx->stream));
        int64_t maxlen = 1; for (int g = 0; g < N; g++) maxlen = std::max<int64_t>(maxlen, gs.lens[(size_t)g]);
        int pos_bits = 1; while (pos_bits < 32 && (1LL << pos_bits) <= maxlen) pos_bits++;
        // the bits above the position hold the first component (0 .. N-1; N = dropped).  An N-way search (mask = every genome) only
        // has matches that start in genome 0: one bit tells them from the dropped ones, which at bacterial sizes saves a sort pass
        const uint32_t full_mask = N >= 32 ? 0xffffffffu : ((1u << N) - 1);
        const bool nway_only = mask != 0 && (uint32_t)mask == full_mask && mode != MAUVE_MODE_PAIRWISE;
        int fbits = 1; if (!nway_only) while ((1 << fbits) <= N) fbits++;
        { KernelTimer t(ctx, MAUVE_K_CANON, ncand);
          hipLaunchKernelGGL(canon_keys, dim3((ncand + 255) / 256), dim3(256), 0, ctx->stream, ctx->mlen.as<int32_t>(),
                             ctx->mstart.as<int32_t>(), ncand, N, pos_bits, nway_only ? 1 : N, ck, cv, ctx->counters.as<uint32_t>() + 3); }
        HIPCHK(ctx, hipGetLastError());
        int rc2 = sort_pairs<uint64_t>(ctx, ncand, pos_bits + fbits, &ck, &cv, ck2, cv2, false, MAUVE_K_CANON);
        if (rc2) return rc2;
        HIPCHK(ctx, ctx->sorted_rec.ensure((size_t)ncand * (1 + N) * 8 + 64));
        hipLaunchKernelGGL(canon_gather, dim3((ncand + 255) / 256), dim3(256), 0, ctx->stream, ctx->mlen.as<int32_t>(),
                           ctx->mstart.as<int32_t>(), ck, cv, ctx->counters.as<uint32_t>() + 3, N, ctx->sorted_rec.as<int64_t>(),
                           ctx->counters.as<uint32_t>() + 4);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, ctx->pin_seed.ensure(64));
        HIPCHK(ctx, hipMemcpyAsync(ctx->pin_seed.p, ctx->counters.p, 32, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        const uint32_t nm = ctx->pin_seed.as<uint32_t>()[3];
        const bool dev_ties = ctx->pin_seed.as<uint32_t>()[4] != 0;
        if (ctx->lazy_matches_ok && nm && !dev_ties) {
            // the caller keeps working on the device copy (sorted_rec); the host copy is made when somebody asks for it
            ctx->match_len.clear(); ctx->match_start.clear();
            ctx->matches_pending = true; ctx->match_nseq = N;
            ctx->n_matches = nm; ctx->dev_rec_n = (int64_t)nm;
            if (n_matches) *n_matches = nm;
            TRACE(ctx, "canonical sort (device, list stays)");
            return MAUVE_OK;
        }
        ctx->match_len.resize(nm); ctx->match_start.resize((size_t)nm * N);
        bool canon_ties = false;
        if (nm) {
            int64_t *ol = ctx->sorted_rec.as<int64_t>();
            // through page-locked staging: a pageable destination of tens of MB copies at a fraction of the link rate
            const size_t rbytes = (size_t)nm * (1 + N) * 8;
            HIPCHK(ctx, ctx->pin_seed.ensure(64 + rbytes));
            char *pin = ctx->pin_seed.as<char>() + 64;
            HIPCHK(ctx, hipMemcpyAsync(pin, ol, rbytes, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            memcpy(ctx->match_len.data(), pin, (size_t)nm * 8);
            memcpy(ctx->match_start.data(), pin + (size_t)nm * 8, (size_t)nm * N * 8);
            // equal (first component, start) groups: order by the rest of the comparator (rare)
            auto k1of = [&](uint32_t r) {
                const int64_t *st = &ctx->match_start[(size_t)r * N];
                int f = 0; while (f < N && st[f] == 0) f++;
                return ((uint64_t)f << 32) | (uint64_t)(f < N ? std::llabs(st[f]) : 0);
            };
            auto rec_less = [&](const std::vector<int64_t> &x, const std::vector<int64_t> &y) {     // [len, starts...]
                uint32_t ma = 0, mb = 0;
                for (int g = 0; g < N; g++) { if (x[1 + g]) ma |= 1u << g; if (y[1 + g]) mb |= 1u << g; }
                if (ma != mb) return ma < mb;
                for (int g = 0; g < N; g++) if (x[1 + g] != y[1 + g]) return x[1 + g] < y[1 + g];
                return x[0] < y[0];
            };
            uint64_t prev = k1of(0);
            for (uint32_t i = 0; i < nm;) {
                uint32_t j = i + 1; uint64_t kj = 0;
                while (j < nm && (kj = k1of(j)) == prev) j++;
                if (j - i > 1) {
                    canon_ties = true;
                    std::vector<std::vector<int64_t>> grp;
                    for (uint32_t r = i; r < j; r++) {
                        std::vector<int64_t> rec(1 + N); rec[0] = ctx->match_len[r];
                        std::copy(&ctx->match_start[(size_t)r * N], &ctx->match_start[(size_t)r * N] + N, rec.begin() + 1);
                        grp.push_back(rec);
                    }
                    std::sort(grp.begin(), grp.end(), rec_less);
                    for (uint32_t r = i; r < j; r++) {
                        ctx->match_len[r] = grp[r - i][0];
                        std::copy(grp[r - i].begin() + 1, grp[r - i].end(), &ctx->match_start[(size_t)r * N]);
                    }
                }
                prev = kj; i = j;
            }
        }
        ctx->n_matches = nm;
        if (canon_ties && nm) {
            // the host finished the order inside the tie groups: the device copy follows (the chaining stages read it)
            char *pin = ctx->pin_seed.as<char>() + 64;
            memcpy(pin, ctx->match_len.data(), (size_t)nm * 8);
            memcpy(pin + (size_t)nm * 8, ctx->match_start.data(), (size_t)nm * N * 8);
            HIPCHK(ctx, hipMemcpyAsync(ctx->sorted_rec.p, pin, (size_t)nm * (1 + N) * 8, hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        }
        ctx->dev_rec_n = (int64_t)nm;                          // sorted_rec holds the list in canonical order
        if (n_matches) *n_matches = nm;
        TRACE(ctx, "canonical sort (device)");
        return MAUVE_OK;
    }
    // small sets: records to the host (page-locked staging), host sort
    if (!records_on_host) {
        hl.resize(ncand); hs.resize((size_t)ncand * N);
        const size_t lbytes = ((size_t)ncand * 4 + 63) & ~(size_t)63, sbytes = (size_t)ncand * 4 * N;
        HIPCHK(ctx, ctx->pin_seed.ensure(64 + lbytes + sbytes));
        char *pin = ctx->pin_seed.as<char>() + 64;
        HIPCHK(ctx, hipMemcpyAsync(pin, ctx->mlen.p, (size_t)ncand * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(pin + lbytes, ctx->mstart.p, sbytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(hl.data(), pin, (size_t)ncand * 4);
        memcpy(hs.data(), pin + lbytes, sbytes);
        TRACE(ctx, "records copy");
    }
    std::vector<uint32_t> &order = ctx->sdh.order; order.clear(); order.reserve(ncand);
    std::vector<uint64_t> &k1 = ctx->sdh.k1; k1.resize(ncand);   // (first component, |start|) packed for a fast first-level compare
    for (uint32_t i = 0; i < ncand; i++) {
        if (hl[i] == 0) continue;
        order.push_back(i);
        const int32_t *s = &hs[(size_t)i * N];
        int f = 0; while (f < N && s[f] == 0) f++;
        uint64_t a = f < N ? (uint64_t)std::abs((int64_t)s[f]) : 0;
        k1[i] = ((uint64_t)f << 40) | a;
    }
    const uint32_t nm = (uint32_t)order.size();
    {   // LSD radix sort of the record indices by k1 (first component << 40 | start): 4 passes of 12 bits,
        // then the rare equal-k1 groups are ordered with the full comparator
        std::vector<uint32_t> &tmp = ctx->sdh.tmp; tmp.resize(nm);
        uint32_t *src = order.data(), *dst = tmp.data();
        for (int pass = 0; pass < 4; pass++) {
            const int sh = 12 * pass;
            uint32_t cnt[4097] = {0};
            for (uint32_t i = 0; i < nm; i++) cnt[(((k1[src[i]] & 0xffffffffULL) | ((k1[src[i]] >> 40) << 32)) >> sh & 4095) + 1]++;
            for (int b = 0; b < 4096; b++) cnt[b + 1] += cnt[b];
            for (uint32_t i = 0; i < nm; i++) dst[cnt[((k1[src[i]] & 0xffffffffULL) | ((k1[src[i]] >> 40) << 32)) >> sh & 4095]++] = src[i];
            std::swap(src, dst);
        }
        if (src != order.data()) std::copy(src, src + nm, order.data());
        auto full_less = [&](uint32_t x, uint32_t y) {
            const int32_t *a = &hs[(size_t)x * N], *b = &hs[(size_t)y * N];
            uint32_t ma = 0, mb = 0;
            for (int g = 0; g < N; g++) { if (a[g]) ma |= 1u << g; if (b[g]) mb |= 1u << g; }
            if (ma != mb) return ma < mb;
            for (int g = 0; g < N; g++) if (a[g] != b[g]) return a[g] < b[g];
            return hl[x] < hl[y];
        };
        for (uint32_t i = 0; i < nm;) {
            uint32_t j = i + 1;
            while (j < nm && k1[order[j]] == k1[order[i]]) j++;
            if (j - i > 1) std::sort(order.begin() + i, order.begin() + j, full_less);
            i = j;
        }
    }
    ctx->match_len.resize(nm); ctx->match_start.resize((size_t)nm * N);
    for (uint32_t i = 0; i < nm; i++) {
        uint32_t o = order[i];
        ctx->match_len[i] = hl[o];
        for (int g = 0; g < N; g++) ctx->match_start[(size_t)i * N + g] = hs[(size_t)o * N + g];
    }
    ctx->n_matches = nm;
    if (n_matches) *n_matches = nm;
    TRACE(ctx, "canonical sort");
    return MAUVE_OK;
}

// stable LSD radix sort of (32-bit key, 32-bit value) pairs for the other translation units (chain_dev.hip)
int sort_pairs_u32(mauve_ctx *ctx, uint32_t n, int key_bits, uint32_t **keys_io, uint32_t **vals_io, uint32_t *keys_alt, uint32_t *vals_alt,
                   int timer_id)
{
    return sort_pairs<uint32_t>(ctx, n, key_bits, keys_io, vals_io, keys_alt, vals_alt, false, timer_id);
}

int seedpass_run(mauve_ctx *ctx, const GenomeSet &gs, uint64_t pattern, int mode, uint64_t mask, int extend,
                 const uint32_t *seg_dev, uint32_t nseg, int64_t *n_matches)
{
    SeedShape sh;
    if (!make_seed_shape(pattern, &sh)) { ctx->err = "seed pattern must be palindromic, span <= 49, weight <= 31"; return MAUVE_ERR_ARG; }
    if (gs.nseq < 1) { ctx->err = "no genomes set"; return MAUVE_ERR_STATE; }
    GenomeTab tab; int64_t total = 0;
    int rc = build_tab(ctx, gs, sh.span, &tab, &total);
    if (rc) return rc;
    ctx->n_matches = 0; ctx->match_len.clear(); ctx->match_start.clear(); ctx->matches_pending = false;
    ctx->dev_rec_n = -1;
    if (n_matches) *n_matches = 0;
    if (total == 0) return MAUVE_OK;
    if (seg_dev) return seedpass_impl<uint64_t, true>(ctx, gs, sh, tab, total, mode, mask, extend, -1, seg_dev, nseg, n_matches, nullptr, nullptr);
    if (2 * sh.weight <= 32) return seedpass_impl<uint32_t, false>(ctx, gs, sh, tab, total, mode, mask, extend, -1, nullptr, 0, n_matches, nullptr, nullptr);
    return seedpass_impl<uint64_t, false>(ctx, gs, sh, tab, total, mode, mask, extend, -1, nullptr, 0, n_matches, nullptr, nullptr);
}

// extension + canonical order of hits a host-side finder supplies (records of 1 + nseq words: component set, values)
int seedpass_from_hits(mauve_ctx *ctx, const GenomeSet &gs, uint64_t pattern, const HostHits &hits, int extend, int64_t *n_matches)
{
    SeedShape sh;
    if (!make_seed_shape(pattern, &sh)) { ctx->err = "seed pattern must be palindromic, span <= 49, weight <= 31"; return MAUVE_ERR_ARG; }
    if (gs.nseq < 1) { ctx->err = "no genomes set"; return MAUVE_ERR_STATE; }
    GenomeTab tab; int64_t total = 0;
    int rc = build_tab(ctx, gs, sh.span, &tab, &total);
    if (rc) return rc;
    ctx->n_matches = 0; ctx->match_len.clear(); ctx->match_start.clear(); ctx->matches_pending = false;
    ctx->dev_rec_n = -1;
    if (n_matches) *n_matches = 0;
    if (total == 0 || hits.n == 0) return MAUVE_OK;
    ctx->host_hits = &hits;
    rc = seedpass_impl<uint32_t, false>(ctx, gs, sh, tab, total, MAUVE_MODE_MEM, 0, extend, -1, nullptr, 0, n_matches, nullptr, nullptr);
    ctx->host_hits = nullptr;
    return rc;
}

// SeedMatchEnumerator::FindMatches of sequence `seq` on the device; q carries the rule in and the CSR result out
int seedpass_enumerate(mauve_ctx *ctx, const GenomeSet &gs, int seq, uint64_t pattern, EnumRequest &q)
{
    SeedShape sh;
    if (!make_seed_shape(pattern, &sh)) { ctx->err = "seed pattern must be palindromic, span <= 49, weight <= 31"; return MAUVE_ERR_ARG; }
    if (seq < 0 || seq >= gs.nseq) { ctx->err = "sequence index out of range"; return MAUVE_ERR_ARG; }
    GenomeTab tab; int64_t total = 0;
    int rc = build_tab(ctx, gs, sh.span, &tab, &total);
    if (rc) return rc;
    q.n = 0; q.ns = 0;
    if (tab.nwin[seq] == 0) { if (q.start_off) q.start_off[0] = 0; return MAUVE_OK; }
    ctx->enum_req = &q;
    if (2 * sh.weight <= 32) rc = seedpass_impl<uint32_t, false>(ctx, gs, sh, tab, total, 0, 0, 0, seq, nullptr, 0, nullptr, nullptr, nullptr);
    else rc = seedpass_impl<uint64_t, false>(ctx, gs, sh, tab, total, 0, 0, 0, seq, nullptr, 0, nullptr, nullptr, nullptr);
    ctx->enum_req = nullptr;
    return rc;
}

int seedpass_sorted_list(mauve_ctx *ctx, const GenomeSet &gs, int seq, uint64_t pattern, std::vector<uint64_t> *keys,
                         std::vector<uint32_t> *vals, int *weight)
{
    SeedShape sh;
    if (!make_seed_shape(pattern, &sh)) { ctx->err = "seed pattern must be palindromic, span <= 49, weight <= 31"; return MAUVE_ERR_ARG; }
    if (seq < 0 || seq >= gs.nseq) { ctx->err = "sequence index out of range"; return MAUVE_ERR_ARG; }
    GenomeTab tab; int64_t total = 0;
    int rc = build_tab(ctx, gs, sh.span, &tab, &total);
    if (rc) return rc;
    *weight = sh.weight;
    keys->clear(); vals->clear();
    if (tab.nwin[seq] == 0) return MAUVE_OK;
    if (2 * sh.weight <= 32) rc = seedpass_impl<uint32_t, false>(ctx, gs, sh, tab, total, 0, 0, 0, seq, nullptr, 0, nullptr, keys, vals);
    else rc = seedpass_impl<uint64_t, false>(ctx, gs, sh, tab, total, 0, 0, 0, seq, nullptr, 0, nullptr, keys, vals);
    if (rc) return rc;
    // vals carry global window indices; make them local to the genome
    for (auto &v : *vals) v = ((v & 0x7fffffffu) - tab.gpos_off[seq]) | (v & 0x80000000u);
    return MAUVE_OK;
}

// host copy of a match list the seed pass left on the device only (seedpass_impl with lazy_matches_ok)
int seed_matches_to_host(mauve_ctx *ctx)
{
    if (!ctx->matches_pending) return MAUVE_OK;
    const size_t nm = (size_t)ctx->n_matches; const int N = ctx->match_nseq;
    const size_t rbytes = nm * (1 + (size_t)N) * 8;
    HIPCHK(ctx, ctx->pin_seed.ensure(64 + rbytes));
    char *pin = ctx->pin_seed.as<char>() + 64;
    HIPCHK(ctx, hipMemcpyAsync(pin, ctx->sorted_rec.p, rbytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->match_len.resize(nm); ctx->match_start.resize(nm * N);
    memcpy(ctx->match_len.data(), pin, nm * 8);
    memcpy(ctx->match_start.data(), pin + nm * 8, nm * N * 8);
    ctx->matches_pending = false;
    return MAUVE_OK;
}
