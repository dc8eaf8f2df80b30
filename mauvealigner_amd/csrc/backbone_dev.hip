// backbone_dev.hip -- backbone segments and pairwise islands of an alignment: the stage that stands in
// for libMems' detectBackbone(iv_list, bb_list, &BigGapsDetector(island_gap_size)) (progressiveMauve.cpp:242-243) and
// for simpleFindIslands (mauveAligner.cpp:844).  The alignment is the column array the assembly stage left in HBM
// (one uint32 presence mask per column); everything that touches columns runs on the device:
//   bb_pair_gaps  one wave (= one workgroup) per (4096-column chunk, genome pair): the pair's gap regions that start in the chunk are
//                 walked 64 columns at a time on wave ballots (all bookkeeping is wave-uniform, i.e. scalar), and the
//                 open regions and the islands are appended to a record list;
//   bb_tile_count per-genome residue counts of every 4096-column tile;  bb_rank  residue counts at query columns
//                 (a wave per query, inside one tile) -- sequence coordinates of the segment and island ends.
// The host only sweeps the region boundaries (10^3..10^4 records) into components and segments.
#include "common.hpp"
#include <algorithm>
#include <cstring>
#include <cmath>

namespace {

constexpr int BB_CHUNK = 4096;            // columns per chunk / per count tile

struct BbIv { int64_t col0; int64_t ncols; uint32_t gmask; uint32_t npairs; uint32_t iv; uint32_t chunk0; };
struct BbRec { uint32_t iv, packed, c_first, c_last; };       // packed: a | b << 8 | who << 16 | kind << 24 (0 region, 1 island)

__device__ __forceinline__ uint64_t below(int p) { return p >= 64 ? ~0ull : (1ull << p) - 1; }

// does m hold a run of at least `need` consecutive ones?  (AND with itself shifted, doubling the length known so far)
__device__ __forceinline__ bool has_run(uint64_t m, int need)
{
    if (need > 64) return false;
    int have = 1;
    while (have < need && m) { const int sh = have < need - have ? have : need - have; m &= m >> sh; have += sh; }
    return m != 0;
}

// the k-th genome pair (a < b) among the genomes of gmask, in the order (g0,g1), (g0,g2), .., (g1,g2), ..
__device__ __forceinline__ void bb_pair_of(uint32_t gmask, uint32_t p, int *a, int *b)
{
    int g[32], n = 0;
    for (int x = 0; x < 32; x++) if (gmask >> x & 1) g[n++] = x;
    int x = 0;
    while (p >= (uint32_t)(n - 1 - x)) { p -= (uint32_t)(n - 1 - x); x++; }
    *a = g[x]; *b = g[x + 1 + (int)p];
}

// what the last pair-column of every (chunk, pair) is: 0 none in the chunk, 1 both genomes, 2 one of them.  bb_pair_gaps finds the
// state in front of its chunk from these bytes, 64 chunks per step, instead of walking back over the columns themselves (a pair
// that is absent over a long stretch of an interval would make every chunk of the stretch walk back to its start)
__global__ void __launch_bounds__(64) bb_chunk_last(const uint32_t *__restrict__ cols, const BbIv *__restrict__ ivs, uint32_t n_ivs, uint32_t max_pairs, uint8_t *__restrict__ last)
{
    const int lane = threadIdx.x & 63;
    uint32_t lo = 0, hi = n_ivs;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (ivs[mid].chunk0 <= blockIdx.x) lo = mid; else hi = mid; }
    const BbIv d = ivs[lo];
    const int64_t nc = d.ncols, cs = (int64_t)(blockIdx.x - d.chunk0) * BB_CHUNK, ce = cs + BB_CHUNK < nc ? cs + BB_CHUNK : nc;
    const uint32_t *m = cols + d.col0;
    // one wave per chunk: the last column of every genome in the chunk (lane g keeps genome g's; the chunk's last 64 columns settle nearly
    // all of them, the rest walk further back), then a pair's last column is the later of its two genomes' -- both if they coincide
    int64_t lastpos = -1;
    uint32_t pending = d.gmask;
    for (int64_t w = (ce - 1) & ~(int64_t)63; w >= cs && pending; w -= 64) {
        const int64_t c = w + lane;
        const uint32_t v = c < ce ? m[c] : 0u;
        for (uint32_t mg = pending; mg; mg &= mg - 1) {
            const int g = __ffs(mg) - 1;
            const uint64_t B = __ballot(v >> g & 1u);
            if (B) { if (lane == g) lastpos = w + 63 - __clzll((long long)B); pending &= ~(1u << g); }
        }
    }
    uint32_t p = 0;
    for (uint32_t ma = d.gmask; ma; ma &= ma - 1)                // pairs in bb_pair_of's order: (g0,g1), (g0,g2), .., (g1,g2), ..
        for (uint32_t mb = ma & (ma - 1); mb; mb &= mb - 1, p++) {
            const int64_t la = __shfl(lastpos, __ffs(ma) - 1, 64), lb = __shfl(lastpos, __ffs(mb) - 1, 64);
            if (lane == 0) last[(size_t)blockIdx.x * max_pairs + p] = (la < 0 && lb < 0) ? 0 : (la == lb ? 1 : 2);
        }
}

__global__ void __launch_bounds__(64) bb_pair_gaps(const uint32_t *__restrict__ cols, const BbIv *__restrict__ ivs, uint32_t n_ivs, uint32_t island_gap,
                                                    BbRec *__restrict__ rec, uint32_t cap, uint32_t *__restrict__ count, const uint8_t *__restrict__ last, uint32_t max_pairs)
{
    const int lane = threadIdx.x & 63;
    // which interval this chunk belongs to
    uint32_t lo = 0, hi = n_ivs;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (ivs[mid].chunk0 <= blockIdx.x) lo = mid; else hi = mid; }
    const BbIv d = ivs[lo];
    const int64_t nc = d.ncols, cs = (int64_t)(blockIdx.x - d.chunk0) * BB_CHUNK, ce = cs + BB_CHUNK < nc ? cs + BB_CHUNK : nc;
    const uint32_t *m = cols + d.col0;
    for (uint32_t p = blockIdx.y; p < d.npairs; p += gridDim.y) {
        int a, b;
        bb_pair_of(d.gmask, p, &a, &b);
        // the nearest column before the chunk that holds a residue of the pair: a both-column (or none) means a region that
        // starts with the chunk's first column is this chunk's to report; a one-sided column means it began earlier
        bool seen_both = false, skipping = false;
        if (!last) {                                     // A/B (MAUVE_BB_COLUMN_SCAN): walk back over the columns themselves
            for (int64_t w = cs - 64; w > -64; w -= 64) {
                const int64_t c = w + lane;
                const uint32_t v = c >= 0 ? m[c] : 0u;
                const bool ra = v >> a & 1, rb = v >> b & 1;
                const uint64_t any = __ballot(ra || rb), both = __ballot(ra && rb);
                if (any) { const int top = 63 - __clzll((long long)any); if (both >> top & 1) seen_both = true; else skipping = true; break; }
            }
        } else
        for (int64_t k0 = (int64_t)blockIdx.x - 1; k0 >= (int64_t)d.chunk0; k0 -= 64) {       // the chunks in front of this one, nearest first, 64 per step
            const int64_t k = k0 - lane;
            const uint32_t stv = k >= (int64_t)d.chunk0 ? last[(size_t)k * max_pairs + p] : 0u;
            const uint64_t any = __ballot(stv != 0);
            if (any) { const int near = __ffsll((long long)any) - 1; if (__shfl((int)stv, near, 64) == 1) seen_both = true; else skipping = true; break; }
        }
        bool in_region = false, leading = false, has_island = false, done = false;
        int run_t = 0;                                   // 1: only a has residues, 2: only b
        int64_t first = 0, last = 0, run_first = 0, run_last = 0, run_n = 0;
        auto emit = [&](uint32_t kind, uint32_t who, int64_t c0, int64_t c1) {
            if (lane == 0) {
                const uint32_t k = atomicAdd(count, 1u);
                if (k < cap) rec[k] = BbRec{d.iv, (uint32_t)a | (uint32_t)b << 8 | who << 16 | kind << 24, (uint32_t)c0, (uint32_t)c1};
            }
        };
        auto close_run = [&]() {
            if (run_n > (int64_t)island_gap) { has_island = true; emit(1u, (uint32_t)(run_t == 1 ? a : b), run_first, run_last); }
        };
        // four words (256 columns) are fetched at a time, one batch ahead; the walk itself stays word by word
        uint32_t nx[4];                                          // the next four words are on their way while these four are walked
#pragma unroll
        for (int k = 0; k < 4; k++) { const int64_t c = cs + 64 * k + lane; nx[k] = c < nc ? m[c] : 0u; }
        for (int64_t w0 = cs; w0 < nc && !done && (w0 < ce || in_region); w0 += 256) {
          uint32_t vv[4];
#pragma unroll
          for (int k = 0; k < 4; k++) { vv[k] = nx[k]; const int64_t c = w0 + 256 + 64 * k + lane; nx[k] = c < nc ? m[c] : 0u; }
          if (!in_region && !skipping) {                         // the common batch: 256 columns without a one-sided one, no region open
              bool one = false, both = false;
#pragma unroll
              for (int k = 0; k < 4; k++) { const bool ra = vv[k] >> a & 1, rb = vv[k] >> b & 1; one |= ra != rb; both |= ra && rb; }
              if (!__ballot(one)) { if (__ballot(both)) seen_both = true; continue; }
          }
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int64_t w = w0 + 64 * k;
            if (!(w < nc && !done && (w < ce || in_region))) break;
            const uint32_t v = vv[k];
            const bool ra = v >> a & 1, rb = v >> b & 1;
            const uint64_t mOne = __ballot(ra != rb);
            if (!mOne && !in_region && !skipping) {              // the common word: nothing one-sided in it and no region open
                if (__ballot(ra && rb)) seen_both = true;
                continue;
            }
            const uint64_t mA = __ballot(ra && !rb), mB = mOne & ~mA, mBoth = __ballot(ra && rb);
            if (!in_region && !skipping) {
                // the usual gapped word: every column holds a base of the pair, the gaps open after a both-column and close
                // before the word's last column, and none of them is an island -- nothing to report, nothing carried over
                const uint64_t valid = __ballot(w + lane < nc);
                const int top = 63 - __clzll((long long)valid);
                if ((mA | mB | mBoth) == valid && (mBoth >> top & 1) && (seen_both || (mBoth & 1)) &&
                    !has_run(mA, (int)island_gap + 1) && !has_run(mB, (int)island_gap + 1)) { seen_both = true; continue; }
            }
            int pos = 0;
            while (pos < 64) {
                if (skipping) {
                    const uint64_t r = mBoth & ~below(pos);
                    if (!r) break;
                    skipping = false; seen_both = true; pos = __ffsll((long long)r);       // the column after the both-column
                    continue;
                }
                if (!in_region) {
                    const uint64_t r = (mA | mB) & ~below(pos);
                    if (!r) { if (mBoth & ~below(pos)) seen_both = true; break; }
                    const int s = __ffsll((long long)r) - 1;
                    if (mBoth & ~below(pos) & below(s)) seen_both = true;
                    if (w + s >= ce) { done = true; break; }             // starts in the next chunk
                    in_region = true; leading = !seen_both; has_island = false;
                    first = last = run_first = run_last = w + s; run_n = 0; run_t = (mA >> s & 1) ? 1 : 2;
                    pos = s;
                }
                // inside a region: one-sided columns up to the next both-column
                const uint64_t rb2 = mBoth & ~below(pos);
                const int q = rb2 ? __ffsll((long long)rb2) - 1 : 64;
                const uint64_t range = below(q) & ~below(pos);
                uint64_t ba = mA & range, bbits = mB & range;
                while (ba | bbits) {
                    const uint64_t same = run_t == 1 ? ba : bbits, other = run_t == 1 ? bbits : ba;
                    const int e = other ? __ffsll((long long)other) - 1 : 64;
                    const uint64_t take = same & below(e);
                    if (take) { run_n += __popcll(take); run_last = w + 63 - __clzll((long long)take); last = run_last; }
                    ba &= ~below(e); bbits &= ~below(e);
                    if (other) { close_run(); run_t = 3 - run_t; run_first = w + e; run_n = 0; }
                }
                if (q < 64) {                                            // the region ends before a both-column
                    close_run();
                    if (has_island || leading) emit(0u, 0u, first, last);
                    in_region = false; seen_both = true; pos = q + 1;
                    if (w + q >= ce) { done = true; break; }
                } else break;
            }
          }
        }
        if (in_region) { close_run(); emit(0u, 0u, first, last); }       // ran into the end of the interval
    }
}

// residues of every genome in every 4096-column tile of the whole column array
__global__ void __launch_bounds__(256) bb_tile_count(const uint32_t *__restrict__ cols, int64_t n, int N, uint32_t *__restrict__ tile_cnt)
{
    __shared__ uint32_t acc[32];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 32) acc[threadIdx.x] = 0;
    __syncthreads();
    const int64_t t0 = (int64_t)blockIdx.x * BB_CHUNK;
    uint32_t mine = 0;
    for (int k = 0; k < BB_CHUNK / 256; k++) {
        const int64_t c = t0 + k * 256 + threadIdx.x;
        const uint32_t v = c < n ? cols[c] : 0u;
        for (int g = 0; g < N; g++) { const uint32_t x = (uint32_t)__popcll(__ballot(v >> g & 1)); if (lane == g) mine += x; }
    }
    if (lane < N) atomicAdd(&acc[lane], mine);
    __syncthreads();
    if ((int)threadIdx.x < N) tile_cnt[(size_t)blockIdx.x * N + threadIdx.x] = acc[threadIdx.x];
}

// residues of every genome in every interval of a caller's column array (check_columns): a tile inside one interval adds its ballot
// counts once, a tile that spans interval borders adds per column
__global__ void __launch_bounds__(256) bb_iv_residues(const uint32_t *__restrict__ cols, int64_t n, int N, const int64_t *__restrict__ off, int64_t n_iv,
                                                      unsigned long long *__restrict__ cnt, uint32_t *__restrict__ bad)
{
    auto iv_of = [&](int64_t c) {                              // the last interval whose first column is <= c (empty intervals are skipped over)
        int64_t lo = 0, hi = n_iv;                             // off[lo] <= c < off[hi]
        while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (off[mid] <= c) lo = mid; else hi = mid; }
        return lo;
    };
    const int lane = threadIdx.x & 63;
    const int64_t t0 = (int64_t)blockIdx.x * BB_CHUNK, t1 = min(n, t0 + (int64_t)BB_CHUNK);
    if (t0 >= t1) return;
    const int64_t iv0 = iv_of(t0), iv1 = iv_of(t1 - 1);
    const uint32_t above = N >= 32 ? 0u : ~0u << N;
    uint32_t mine = 0; bool wrong = false;
    for (int k = 0; k < BB_CHUNK / 256; k++) {
        const int64_t c = t0 + k * 256 + threadIdx.x;
        const uint32_t v = c < t1 ? cols[c] : 0u;
        wrong |= (v & above) != 0;
        if (iv0 == iv1) {
            for (int g = 0; g < N; g++) { const uint32_t x = (uint32_t)__popcll(__ballot(v >> g & 1)); if (lane == g) mine += x; }
        } else if (v) {
            const int64_t iv = iv_of(c);
            for (uint32_t m = v & ~above; m; m &= m - 1) atomicAdd(&cnt[(size_t)iv * N + (__ffs(m) - 1)], 1ull);
        }
    }
    if (iv0 == iv1 && lane < N && mine) atomicAdd(&cnt[(size_t)iv0 * N + lane], (unsigned long long)mine);
    if (wrong) atomicOr(bad, 1u);
}

// residues of every genome in [tile start of x, x) for every query column x: the tile's 64 words are split over the four
// waves of the workgroup, four words in flight per lane
__global__ void __launch_bounds__(256) bb_rank(const uint32_t *__restrict__ cols, const int64_t *__restrict__ query, int N, uint32_t *__restrict__ out)
{
    __shared__ uint32_t acc[32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 32) acc[threadIdx.x] = 0;
    __syncthreads();
    const int64_t x = query[blockIdx.x], t0 = (x & ~(int64_t)(BB_CHUNK - 1)) + (int64_t)wave * (BB_CHUNK / 4);
    uint32_t mine = 0;
    for (int64_t w = t0; w < x && w < t0 + BB_CHUNK / 4; w += 256) {
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { const int64_t c = w + 64 * k + lane; v[k] = c < x ? cols[c] : 0u; }
#pragma unroll
        for (int k = 0; k < 4; k++)
            for (int g = 0; g < N; g++) { const uint32_t n = (uint32_t)__popcll(__ballot(v[k] >> g & 1)); if (lane == g) mine += n; }
    }
    if (lane < N && mine) atomicAdd(&acc[lane], mine);
    __syncthreads();
    if ((int)threadIdx.x < N) out[(size_t)blockIdx.x * N + threadIdx.x] = acc[threadIdx.x];
}

// connected components (>= 2 genomes) of the joined pairs: adj[g] = genomes joined to g (bit g included)
void bb_components(const uint32_t *adj, uint32_t gmask, std::vector<uint32_t> &out)
{
    out.clear();
    uint32_t left = gmask;
    while (left) {
        const int g = __builtin_ctz(left);
        uint32_t comp = 1u << g, front = comp;
        while (front) {
            uint32_t nxt = 0;
            for (uint32_t f = front; f; f &= f - 1) nxt |= adj[__builtin_ctz(f)];
            front = nxt & gmask & ~comp; comp |= front;
        }
        left &= ~comp;
        if (comp & (comp - 1)) out.push_back(comp);
    }
}

// ================================================================================================
// Homology pass: the two-state pair HMM in front of the backbone, on the device.
// With x = vH - vU the Viterbi recurrence of a pair-column with emission score s is ONE clamp:  x' = s + clamp(x, go_h, -go_u)
// (x <= go_h: H is entered from U; x >= -go_u: U is entered from H; in between both stay), the predecessor choices are
// "x >= go_h" (H came from H) and "x <= -go_u" (U came from U), the final state is "x >= 0".  Functions x -> clamp(x + t, L, U)
// are closed under composition, so the columns are an associative scan over (t, L, U) triples of int32 -- and the way back
// (state -> predecessor state, a map on two elements = two bits per column) is one too.  Intervals are cut into segments of
// 4096 columns (64 steps of 64 lanes); every (interval, pair, segment) is one wave:
//   hom_seg_count  residues of every genome per segment (base positions at the segment starts: host prefix per interval)
//   hom_fn         the segment's composed function                      -> host: x at every segment start, final state
//   hom_pred       the same sweep with x known: predecessor words (2 x 64 bits per step), step maps, the segment's map
//                                                                       -> host: state at every segment end
//   hom_keep       states of all columns from the words and the end state; ORs the pair's bits into keep[column] where the
//                  state is H and both genomes have a base
//   hom_count / hom_write / hom_offsets  the columns split by keep: tile totals, (host prefix over the tiles), the new
//                columns written in order, the new column offset of every interval.
constexpr int HOM_SEG = 4096;                     // columns per segment = 64 steps of 64 lanes
struct HomItem { uint32_t ivx, seg; uint8_t a, b; uint16_t pad; };                            // one wave's work: segment `seg` of interval ivx, pair (a, b)
struct HomIv { int64_t col0, ncols; uint32_t iv, nseg; uint32_t seg0, pad; };                  // seg0: the interval's first row in the per-segment tables
struct HomGenomes { uint64_t word_off[MAUVE_MAX_SEQ]; };
struct HomScores { int32_t match, mismatch, gap, go_h, go_u; };
struct HomFn { int32_t t, lo, hi, pad; };                                                     // x -> clamp(x + t, lo, hi)
constexpr int32_t HOM_MINF = -(1 << 30), HOM_PINF = 1 << 30;

__device__ __forceinline__ int hom_base(const uint64_t *__restrict__ G, int64_t pos0) { return (int)(G[pos0 >> 5] >> ((pos0 & 31) * 2) & 3u); }
__device__ __forceinline__ int32_t hom_clamp(int32_t x, int32_t lo, int32_t hi) { return x < lo ? lo : (x > hi ? hi : x); }
// g2 after g1
__device__ __forceinline__ void hom_compose(int32_t &t, int32_t &lo, int32_t &hi, int32_t t1, int32_t lo1, int32_t hi1)
{
    const int32_t t2 = t, lo2 = lo, hi2 = hi;
    t = t1 + t2; lo = hom_clamp(lo1 + t2, lo2, hi2); hi = hom_clamp(hi1 + t2, lo2, hi2);
    if (lo1 == HOM_MINF && lo2 == HOM_MINF) lo = HOM_MINF;      // two identities stay the identity
    if (hi1 == HOM_PINF && hi2 == HOM_PINF) hi = HOM_PINF;
}
// maps on {U = 0, H = 1} as two bits (bit s = image of s); p after q
__device__ __forceinline__ uint32_t hom_map_after(uint32_t p, uint32_t q) { return ((p >> (q & 1u)) & 1u) | (((p >> (q >> 1 & 1u)) & 1u) << 1); }

struct HomPair {          // what a wave knows about its (interval, pair, segment)
    const uint64_t *Ga, *Gb; int64_t la, ra, lb, rb; bool rva, rvb; int a, b; int64_t c0, n; int64_t ka, kb;
};
__device__ __forceinline__ HomPair hom_setup(const HomItem &item, const HomIv &iv, const int64_t *__restrict__ left, const int64_t *__restrict__ right,
                                             const int8_t *__restrict__ rev, int N, const uint64_t *__restrict__ genomes, const HomGenomes &gw,
                                             const int64_t *__restrict__ rank0)
{
    HomPair P;
    P.a = item.a; P.b = item.b;
    P.la = left[(size_t)iv.iv * N + P.a]; P.ra = right[(size_t)iv.iv * N + P.a]; P.lb = left[(size_t)iv.iv * N + P.b]; P.rb = right[(size_t)iv.iv * N + P.b];
    P.rva = rev[(size_t)iv.iv * N + P.a] != 0; P.rvb = rev[(size_t)iv.iv * N + P.b] != 0;
    P.Ga = genomes + gw.word_off[P.a]; P.Gb = genomes + gw.word_off[P.b];
    P.c0 = iv.col0 + (int64_t)item.seg * HOM_SEG;
    P.n = min((int64_t)HOM_SEG, iv.ncols - (int64_t)item.seg * HOM_SEG);
    P.ka = rank0[(size_t)(iv.seg0 + item.seg) * N + P.a]; P.kb = rank0[(size_t)(iv.seg0 + item.seg) * N + P.b];
    return P;
}
// the column of this lane in step `ch` as a function (identity where neither genome has a base); advances the base counters
__device__ __forceinline__ void hom_column(HomPair &P, const uint32_t *__restrict__ cols, int ch, int lane, const HomScores &sc, int32_t &t, int32_t &lo, int32_t &hi,
                                           bool &none, bool &both)
{
    const int64_t c = (int64_t)ch * 64 + lane;
    const uint32_t m = c < P.n ? cols[P.c0 + c] : 0u;
    const bool ha = m >> P.a & 1u, hb = m >> P.b & 1u;
    none = !(ha || hb); both = ha && hb;
    const uint64_t BA = __ballot(ha), BB = __ballot(hb);
    int32_t s = sc.gap;
    if (both) {
        const int64_t ia = P.ka + __popcll(BA & below(lane)), ib = P.kb + __popcll(BB & below(lane));
        int xa = hom_base(P.Ga, (P.rva ? P.ra - ia : P.la + ia) - 1), xb = hom_base(P.Gb, (P.rvb ? P.rb - ib : P.lb + ib) - 1);
        if (P.rva) xa = 3 - xa;
        if (P.rvb) xb = 3 - xb;
        s = xa == xb ? sc.match : sc.mismatch;
    }
    t = none ? 0 : s; lo = none ? HOM_MINF : sc.go_h + s; hi = none ? HOM_PINF : -sc.go_u + s;
    P.ka += __popcll(BA); P.kb += __popcll(BB);
}
// inclusive scan over the lanes: lane l gets (column l) after ... after (column 0)
__device__ __forceinline__ void hom_scan(int lane, int32_t &t, int32_t &lo, int32_t &hi)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t t1 = __shfl_up(t, d, 64), lo1 = __shfl_up(lo, d, 64), hi1 = __shfl_up(hi, d, 64);
        if (lane >= d) hom_compose(t, lo, hi, t1, lo1, hi1);
    }
}

__global__ void __launch_bounds__(64) hom_seg_count(const uint32_t *__restrict__ cols, const HomIv *__restrict__ ivs, uint32_t n_ivs, int N, uint32_t *__restrict__ cnt)
{
    const int lane = threadIdx.x;
    // which interval: the segment rows are consecutive per interval
    uint32_t lo = 0, hi = n_ivs;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (ivs[mid].seg0 <= blockIdx.x) lo = mid; else hi = mid; }
    const HomIv iv = ivs[lo];
    const uint32_t seg = blockIdx.x - iv.seg0;
    const int64_t c0 = iv.col0 + (int64_t)seg * HOM_SEG, n = min((int64_t)HOM_SEG, iv.ncols - (int64_t)seg * HOM_SEG);
    uint32_t mine = 0;
    for (int ch = 0; ch * 64 < n; ch++) {
        const int64_t c = (int64_t)ch * 64 + lane;
        const uint32_t v = c < n ? cols[c0 + c] : 0u;
        for (int g = 0; g < N; g++) { const uint32_t x = (uint32_t)__popcll(__ballot(v >> g & 1)); if (lane == g) mine += x; }
    }
    if (lane < N) cnt[(size_t)blockIdx.x * N + lane] = mine;
}

__global__ void __launch_bounds__(64) hom_fn(const uint32_t *__restrict__ cols, const HomItem *__restrict__ items, uint32_t n_items, const HomIv *__restrict__ ivs,
                                             const int64_t *__restrict__ left, const int64_t *__restrict__ right, const int8_t *__restrict__ rev, int N,
                                             const uint64_t *__restrict__ genomes, HomGenomes gw, HomScores sc, const int64_t *__restrict__ rank0, HomFn *__restrict__ fn)
{
    const int lane = threadIdx.x;
    for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        const HomItem item = items[it];
        HomPair P = hom_setup(item, ivs[item.ivx], left, right, rev, N, genomes, gw, rank0);
        int32_t ft = 0, flo = HOM_MINF, fhi = HOM_PINF;          // the segment so far
        for (int ch = 0; (int64_t)ch * 64 < P.n; ch++) {
            int32_t t, lo, hi; bool none, both;
            hom_column(P, cols, ch, lane, sc, t, lo, hi, none, both);
            hom_scan(lane, t, lo, hi);
            int32_t ct = __shfl(t, 63, 64), clo = __shfl(lo, 63, 64), chi = __shfl(hi, 63, 64);        // the step as a whole ...
            hom_compose(ct, clo, chi, ft, flo, fhi);                                                  // ... after the segment so far
            ft = ct; flo = clo; fhi = chi;
        }
        if (lane == 0) fn[it] = HomFn{ft, flo, fhi, 0};
    }
}

__global__ void __launch_bounds__(64) hom_pred(const uint32_t *__restrict__ cols, const HomItem *__restrict__ items, uint32_t n_items, const HomIv *__restrict__ ivs,
                                               const int64_t *__restrict__ left, const int64_t *__restrict__ right, const int8_t *__restrict__ rev, int N,
                                               const uint64_t *__restrict__ genomes, HomGenomes gw, HomScores sc, const int64_t *__restrict__ rank0,
                                               const int32_t *__restrict__ xin, uint64_t *__restrict__ pred, uint8_t *__restrict__ stepmap, uint8_t *__restrict__ segmap)
{
    const int lane = threadIdx.x;
    for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        const HomItem item = items[it];
        HomPair P = hom_setup(item, ivs[item.ivx], left, right, rev, N, genomes, gw, rank0);
        int32_t x = xin[it];
        uint32_t my_step = 2u;                                  // lane k keeps the map of step k (identity = 0b10)
        for (int ch = 0; (int64_t)ch * 64 < P.n; ch++) {
            int32_t t, lo, hi; bool none, both;
            hom_column(P, cols, ch, lane, sc, t, lo, hi, none, both);
            hom_scan(lane, t, lo, hi);
            const int32_t after = hom_clamp(x + t, lo, hi);     // x behind this lane's column
            int32_t before = __shfl_up(after, 1, 64);
            if (lane == 0) before = x;
            const bool ph = none || before >= sc.go_h, pu = none || before <= -sc.go_u;
            const uint64_t PH = __ballot(ph), PU = __ballot(pu);
            if (lane == 0) { pred[((size_t)it * 64 + ch) * 2] = PH; pred[((size_t)it * 64 + ch) * 2 + 1] = PU; }
            // the way back through this step: m_c(state) = state ? ph : !pu; lanes compose m_l after ... wait: m_l o m_{l+1} o .. o m_63
            uint32_t mp = (ph ? 2u : 0u) | (pu ? 0u : 1u);
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_down(mp, d, 64); if (lane + d < 64) mp = hom_map_after(mp, o); }
            const uint32_t whole = __shfl(mp, 0, 64);           // state behind the step -> state behind the previous step
            if (lane == ch) my_step = whole;
            x = __shfl(after, 63, 64);
        }
        stepmap[(size_t)it * 64 + lane] = (uint8_t)my_step;
        uint32_t sm = my_step;                                  // segment: step 0 after step 1 after ... (state behind the segment -> behind the previous one)
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_down(sm, d, 64); if (lane + d < 64) sm = hom_map_after(sm, o); }
        if (lane == 0) segmap[it] = (uint8_t)sm;
    }
}

__global__ void __launch_bounds__(64) hom_keep(const uint32_t *__restrict__ cols, const HomItem *__restrict__ items, uint32_t n_items, const HomIv *__restrict__ ivs,
                                               const uint64_t *__restrict__ pred, const uint8_t *__restrict__ stepmap, const uint8_t *__restrict__ send,
                                               uint32_t *__restrict__ keep)
{
    const int lane = threadIdx.x;
    for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        const HomItem item = items[it];
        const HomIv iv = ivs[item.ivx];
        const int64_t c0 = iv.col0 + (int64_t)item.seg * HOM_SEG, n = min((int64_t)HOM_SEG, iv.ncols - (int64_t)item.seg * HOM_SEG);
        // state behind every step: lane k = behind step k, from the state behind the segment and the maps of the steps after k
        uint32_t sm = stepmap[(size_t)it * 64 + lane];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_down(sm, d, 64); if (lane + d < 64) sm = hom_map_after(sm, o); }
        const uint32_t e = send[it];                            // state behind the segment
        const uint32_t nxt = __shfl_down(sm, 1, 64);            // steps k+1 .. 63 as one map
        const uint32_t behind = lane == 63 ? e : ((nxt >> e) & 1u);
        for (int ch = 0; (int64_t)ch * 64 < n; ch++) {
            const int64_t c = (int64_t)ch * 64 + lane;
            const uint32_t m = c < n ? cols[c0 + c] : 0u;
            const uint64_t PH = pred[((size_t)it * 64 + ch) * 2], PU = pred[((size_t)it * 64 + ch) * 2 + 1];
            const uint32_t eb = __shfl(behind, ch, 64);         // state of the step's last column... behind the step = state AT its last column
            uint32_t mp = ((PH >> lane & 1ull) ? 2u : 0u) | ((PU >> lane & 1ull) ? 0u : 1u);
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_down(mp, d, 64); if (lane + d < 64) mp = hom_map_after(mp, o); }
            const uint32_t nx = __shfl_down(mp, 1, 64);         // columns l+1 .. 63 as one map: state at column 63 -> state at column l
            const uint32_t st = lane == 63 ? eb : ((nx >> eb) & 1u);
            if (st && (m >> item.a & 1u) && (m >> item.b & 1u)) atomicOr(&keep[c0 + c], 1u << item.a | 1u << item.b);
        }
    }
}

// columns a column becomes: the genomes that stay (one column, if any) + one column per genome that leaves
__device__ __forceinline__ uint32_t hom_out_count(uint32_t m, uint32_t k) { return (k ? 1u : 0u) + (uint32_t)__popc(m & ~k); }

__global__ void __launch_bounds__(256) hom_count(const uint32_t *__restrict__ cols, const uint32_t *__restrict__ keep, int64_t n, uint32_t *__restrict__ tile_out,
                                                 unsigned long long *__restrict__ moved)
{
    __shared__ uint32_t acc[2];
    if (threadIdx.x < 2) acc[threadIdx.x] = 0;
    __syncthreads();
    const int64_t t0 = (int64_t)blockIdx.x * BB_CHUNK;
    uint32_t mine = 0, mv = 0;
    for (int k = 0; k < BB_CHUNK / 256; k++) {
        const int64_t c = t0 + k * 256 + threadIdx.x;
        if (c >= n) break;
        const uint32_t m = cols[c], kp = keep[c] & m;
        mine += hom_out_count(m, kp);
        if (m & (m - 1)) mv += (uint32_t)__popc(m & ~kp);
    }
    atomicAdd(&acc[0], mine); atomicAdd(&acc[1], mv);
    __syncthreads();
    if (threadIdx.x == 0) { tile_out[blockIdx.x] = acc[0]; if (acc[1]) atomicAdd(moved, (unsigned long long)acc[1]); }
}

__global__ void __launch_bounds__(256) hom_write(const uint32_t *__restrict__ cols, const uint32_t *__restrict__ keep, int64_t n, const int64_t *__restrict__ tile_base,
                                                 uint32_t *__restrict__ out)
{
    __shared__ uint32_t wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t base = tile_base[blockIdx.x];
    const int64_t t0 = (int64_t)blockIdx.x * BB_CHUNK;
    for (int k = 0; k < BB_CHUNK / 256; k++) {                // 256 columns per round, in column order
        const int64_t c = t0 + k * 256 + threadIdx.x;
        const uint32_t m = c < n ? cols[c] : 0u, kp = c < n ? keep[c] & m : 0u;
        const uint32_t cnt = c < n ? hom_out_count(m, kp) : 0u;
        uint32_t x = cnt;                                     // inclusive scan over the wave, then over the four waves
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (int w = 0; w < 4; w++) { if (w < wave) before += wsum[w]; total += wsum[w]; }
        int64_t o = base + before + x - cnt;
        if (kp) out[o++] = kp;
        for (uint32_t r = m & ~kp; r; r &= r - 1) out[o++] = r & (0u - r);
        base += total;
        __syncthreads();
    }
}

// new offset of every interval start: the tile's base + the columns the tile puts out before it
__global__ void __launch_bounds__(64) hom_offsets(const uint32_t *__restrict__ cols, const uint32_t *__restrict__ keep, int64_t n, const int64_t *__restrict__ tile_base,
                                                  const int64_t *__restrict__ col_off, int64_t n_off, int64_t *__restrict__ out)
{
    const int lane = threadIdx.x;
    (void)n;
    for (int64_t q = blockIdx.x; q < n_off; q += gridDim.x) {
        const int64_t x = col_off[q], t0 = x & ~(int64_t)(BB_CHUNK - 1);
        uint32_t mine = 0;
        for (int64_t c = t0 + lane; c < x; c += 64) { const uint32_t m = cols[c]; mine += hom_out_count(m, keep[c] & m); }
#pragma unroll
        for (int d = 32; d; d >>= 1) mine += __shfl_xor(mine, d, 64);
        if (lane == 0) out[q] = tile_base[x / BB_CHUNK] + mine;          // (tile_base has one entry behind the last tile: x == n on a tile boundary)
    }
}

}  // namespace

// The work behind mauve_backbone / mauve_backbone_alignment: cols on the device (n_cols entries), interval table on the host.
int backbone_run(mauve_ctx *c, int N, int64_t n_iv, const int64_t *left, const int64_t *right, const int8_t *reverse, const int64_t *col_off,
                 const uint32_t *d_cols, int64_t island_gap)
{
    mauve_ctx::BackboneResult &B = c->bb;
    B = mauve_ctx::BackboneResult(); B.N = N;
    if (island_gap < 0 || island_gap > 0x7fffffff) { c->err = "backbone: island_gap_size out of range"; return MAUVE_ERR_ARG; }
    const int64_t n_cols = col_off[n_iv];
    // intervals with >= 2 genomes, their chunks
    std::vector<BbIv> ivs; uint32_t chunks = 0, max_pairs = 0;
    for (int64_t iv = 0; iv < n_iv; iv++) {
        uint32_t gm = 0; for (int g = 0; g < N; g++) if (left[iv * N + g]) gm |= 1u << g;
        const int n = __builtin_popcount(gm); const int64_t nc = col_off[iv + 1] - col_off[iv];
        if (n < 2 || nc <= 0) continue;
        if (nc > 0x7fffffff) { c->err = "backbone: interval too long"; return MAUVE_ERR_LIMIT; }
        BbIv d{col_off[iv], nc, gm, (uint32_t)(n * (n - 1) / 2), (uint32_t)iv, chunks};
        chunks += (uint32_t)((nc + BB_CHUNK - 1) / BB_CHUNK); max_pairs = std::max(max_pairs, d.npairs);
        ivs.push_back(d);
    }
    if (ivs.empty()) { B.valid = true; return MAUVE_OK; }
    const double tb0 = now_ms();
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n_tiles = (size_t)((n_cols + BB_CHUNK - 1) / BB_CHUNK);
    // work area: interval table | counter | tile counts | records
    auto up = [](size_t x) { return (x + 63) & ~(size_t)63; };
    const size_t o_cnt = up(ivs.size() * sizeof(BbIv)), o_tile = o_cnt + 64, o_last = o_tile + up(n_tiles * (size_t)N * 4), o_rec = o_last + up((size_t)chunks * max_pairs);
    size_t cap = std::max<size_t>(1u << 16, c->bb_rec_cap);
    std::vector<BbRec> recs;
    std::vector<uint32_t> tile_cnt(n_tiles * (size_t)N);
    for (int attempt = 0;; attempt++) {
        HIPCHK(c, c->bb_work.ensure(o_rec + cap * sizeof(BbRec)));
        char *wk = c->bb_work.as<char>();
        HIPCHK(c, hipMemcpyAsync(wk, ivs.data(), ivs.size() * sizeof(BbIv), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemsetAsync(wk + o_cnt, 0, 64, c->stream));
        const uint32_t gy = std::min<uint32_t>(max_pairs, 64);        // one wave per (chunk, pair): with few pairs, too, every resident wave works
        static const bool column_scan = getenv("MAUVE_BB_COLUMN_SCAN") != nullptr;
        // (every attempt: a grown work area is a new allocation)
        if (!column_scan)
            hipLaunchKernelGGL(bb_chunk_last, dim3(chunks), dim3(64), 0, c->stream, d_cols, reinterpret_cast<const BbIv *>(wk), (uint32_t)ivs.size(), max_pairs,
                               reinterpret_cast<uint8_t *>(wk + o_last));
        hipLaunchKernelGGL(bb_pair_gaps, dim3(chunks, gy), dim3(64), 0, c->stream, d_cols, reinterpret_cast<const BbIv *>(wk), (uint32_t)ivs.size(),
                           (uint32_t)island_gap, reinterpret_cast<BbRec *>(wk + o_rec), (uint32_t)cap, reinterpret_cast<uint32_t *>(wk + o_cnt),
                           column_scan ? (const uint8_t *)nullptr : reinterpret_cast<const uint8_t *>(wk + o_last), max_pairs);
        if (attempt == 0)
            hipLaunchKernelGGL(bb_tile_count, dim3((uint32_t)n_tiles), dim3(256), 0, c->stream, d_cols, n_cols, N, reinterpret_cast<uint32_t *>(wk + o_tile));
        HIPCHK(c, hipGetLastError());
        uint32_t n_rec = 0;
        HIPCHK(c, hipMemcpyAsync(&n_rec, wk + o_cnt, 4, hipMemcpyDeviceToHost, c->stream));
        if (attempt == 0) HIPCHK(c, hipMemcpyAsync(tile_cnt.data(), wk + o_tile, tile_cnt.size() * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (n_rec > cap) { cap = (size_t)n_rec + 1024; c->bb_rec_cap = cap; continue; }    // the list did not fit: once more with room
        recs.resize(n_rec);
        if (n_rec) HIPCHK(c, hipMemcpy(recs.data(), wk + o_rec, (size_t)n_rec * sizeof(BbRec), hipMemcpyDeviceToHost));
        break;
    }
    static const bool trace = getenv("MAUVE_TRACE") != nullptr;
    const double tb1 = now_ms();
    if (trace) fprintf(stderr, "[trace] backbone: %u chunks, %zu records; kernels + copies %.3f ms\n", chunks, recs.size(), tb1 - tb0);
    // canonical order of the records: interval, kind, pair, first column (the append order is not deterministic)
    std::sort(recs.begin(), recs.end(), [](const BbRec &x, const BbRec &y) {
        if (x.iv != y.iv) return x.iv < y.iv;
        const uint32_t kx = x.packed >> 24, ky = y.packed >> 24; if (kx != ky) return kx < ky;
        const uint32_t px = (x.packed & 0xff) << 8 | (x.packed >> 8 & 0xff), py = (y.packed & 0xff) << 8 | (y.packed >> 8 & 0xff);
        if (px != py) return px < py;
        return x.c_first < y.c_first;
    });
    // ---- regions -> segments: sweep the boundaries of the open regions, interval by interval
    struct Seg { uint32_t ivx; uint32_t mask; int64_t c1, c2; };
    std::vector<Seg> segs;
    std::vector<BbRec> isl;
    {
        size_t r = 0;
        std::vector<std::pair<int64_t, uint32_t>> ev;              // (column, pair code | open << 31)
        std::vector<uint32_t> comps, prev_comps; std::vector<int64_t> prev_start, start;
        for (size_t x = 0; x < ivs.size(); x++) {
            const BbIv &d = ivs[x];
            ev.clear();
            for (; r < recs.size() && recs[r].iv == d.iv; r++) {
                const BbRec &q = recs[r];
                if (q.packed >> 24) { isl.push_back(q); continue; }
                const uint32_t pc = q.packed & 0xffff;
                ev.push_back({(int64_t)q.c_first, pc | 0x80000000u}); ev.push_back({(int64_t)q.c_last + 1, pc});
            }
            std::sort(ev.begin(), ev.end());
            uint32_t adj[32];
            for (int g = 0; g < 32; g++) adj[g] = d.gmask;             // every pair joined until a region opens
            prev_comps.clear(); prev_start.clear();
            size_t e = 0; int64_t at = 0;
            for (;;) {
                // apply the events at column `at`, then the partition holds from `at` to the next event
                for (; e < ev.size() && ev[e].first == at; e++) {
                    const int a = (int)(ev[e].second & 0xff), b = (int)(ev[e].second >> 8 & 0xff);
                    if (ev[e].second >> 31) { adj[a] &= ~(1u << b); adj[b] &= ~(1u << a); } else { adj[a] |= 1u << b; adj[b] |= 1u << a; }
                }
                if (at >= d.ncols) comps.clear(); else bb_components(adj, d.gmask, comps);
                start.assign(comps.size(), at);
                for (size_t k = 0; k < prev_comps.size(); k++) {
                    const auto it = std::find(comps.begin(), comps.end(), prev_comps[k]);
                    if (it != comps.end()) start[(size_t)(it - comps.begin())] = prev_start[k];
                    else segs.push_back(Seg{(uint32_t)x, prev_comps[k], prev_start[k], at - 1});
                }
                prev_comps = comps; prev_start = start;
                if (at >= d.ncols) break;
                at = e < ev.size() ? std::min<int64_t>(ev[e].first, d.ncols) : d.ncols;
            }
        }
    }
    const double tb2 = now_ms();
    // ---- residue counts at the segment and island ends.  The query slots are laid out by construction -- interval starts,
    // then two per segment, then two per island -- so nothing has to be sorted or looked up (a column asked twice costs a wave)
    std::vector<int64_t> qcol;
    qcol.reserve(ivs.size() + 2 * (segs.size() + isl.size()));
    for (const BbIv &d : ivs) qcol.push_back(d.col0);
    const size_t q_seg = qcol.size();
    for (const Seg &s : segs) { qcol.push_back(ivs[s.ivx].col0 + s.c1); qcol.push_back(ivs[s.ivx].col0 + s.c2 + 1); }
    const size_t q_isl = qcol.size();
    std::vector<uint32_t> isl_ivx(isl.size());                          // the islands' intervals (records and ivs are both ordered by interval)
    { size_t x = 0; for (size_t k = 0; k < isl.size(); k++) { while (ivs[x].iv != isl[k].iv) x++; isl_ivx[k] = (uint32_t)x; } }
    for (size_t k = 0; k < isl.size(); k++) { const BbIv &d = ivs[isl_ivx[k]]; qcol.push_back(d.col0 + isl[k].c_first); qcol.push_back(d.col0 + (int64_t)isl[k].c_last + 1); }
    // many genomes with many islands ask for the same columns pair after pair: beyond a million queries they are made unique
    // first (one sort) and every slot is looked up once
    std::vector<uint32_t> slot;
    if (qcol.size() > (1u << 20)) {
        std::vector<int64_t> all(qcol);
        std::sort(qcol.begin(), qcol.end()); qcol.erase(std::unique(qcol.begin(), qcol.end()), qcol.end());
        slot.resize(all.size());
        for (size_t k = 0; k < all.size(); k++) slot[k] = (uint32_t)(std::lower_bound(qcol.begin(), qcol.end(), all[k]) - qcol.begin());
    }
    const uint32_t *qcnt;
    {
        const size_t o_q = 0, o_out = up(qcol.size() * 8), n_out = qcol.size() * (size_t)N * 4;
        HIPCHK(c, c->bb_query.ensure(o_out + n_out + 64));
        HIPCHK(c, c->pin_bb.ensure(o_out + n_out + 64));                // page-locked staging, same layout
        char *qb = c->bb_query.as<char>(), *hb = c->pin_bb.as<char>();
        memcpy(hb + o_q, qcol.data(), qcol.size() * 8);
        HIPCHK(c, hipMemcpyAsync(qb + o_q, hb + o_q, qcol.size() * 8, hipMemcpyHostToDevice, c->stream));
        for (size_t q0 = 0; q0 < qcol.size(); q0 += (size_t)1 << 22) {      // grid x block stays below 2^32 threads per launch
            const size_t nq = std::min<size_t>((size_t)1 << 22, qcol.size() - q0);
            hipLaunchKernelGGL(bb_rank, dim3((uint32_t)nq), dim3(256), 0, c->stream, d_cols, reinterpret_cast<const int64_t *>(qb + o_q) + q0, N,
                               reinterpret_cast<uint32_t *>(qb + o_out) + q0 * (size_t)N);
        }
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(hb + o_out, qb + o_out, n_out, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        qcnt = reinterpret_cast<const uint32_t *>(hb + o_out);
    }
    const double tb3 = now_ms();
    if (trace) fprintf(stderr, "[trace] backbone: sort + sweep %.3f ms (%zu segments), %zu rank queries %.3f ms\n", tb2 - tb1, segs.size(), qcol.size(), tb3 - tb2);
    // exclusive prefix of the tile counts
    std::vector<int64_t> tile_pre((n_tiles + 1) * (size_t)N, 0);
    for (size_t t = 0; t < n_tiles; t++) for (int g = 0; g < N; g++) tile_pre[(t + 1) * N + g] = tile_pre[t * N + g] + tile_cnt[t * N + g];
    auto count_at = [&](int64_t x, size_t k, int g) {                  // residues of g in columns [0, x) of the whole array; k = the query's slot
        if (!slot.empty()) k = slot[k];
        return tile_pre[(size_t)(x / BB_CHUNK) * N + g] + (int64_t)qcnt[k * N + g];
    };
    size_t kb = 0, ka = 0, kz = 0;                                     // query slots of the interval start and of the two ends in turn
    auto ends = [&](const BbIv &d, int g, int64_t c1, int64_t c2, int64_t *lo, int64_t *hi) {     // signed ends of g's residues in columns [c1, c2]
        const int64_t base = count_at(d.col0, kb, g), k1 = count_at(d.col0 + c1, ka, g) - base, k2 = count_at(d.col0 + c2 + 1, kz, g) - base - 1;
        if (k2 < k1) return false;
        const int64_t L = left[(int64_t)d.iv * N + g], R = right[(int64_t)d.iv * N + g];
        if (!reverse[(int64_t)d.iv * N + g]) { *lo = L + k1; *hi = L + k2; } else { *lo = -(R - k2); *hi = -(R - k1); }
        return true;
    };
    B.seg_iv.reserve(segs.size()); B.seg_col.reserve(segs.size()); B.seg_len.reserve(segs.size()); B.seg_mask.reserve(segs.size());
    B.seg_left.reserve(segs.size() * (size_t)N); B.seg_right.reserve(segs.size() * (size_t)N);
    for (size_t si = 0; si < segs.size(); si++) {
        const Seg &s = segs[si];
        const BbIv &d = ivs[s.ivx];
        kb = s.ivx; ka = q_seg + 2 * si; kz = ka + 1;
        int64_t lo[32] = {0}, hi[32] = {0}; uint32_t got = 0;
        for (int g = 0; g < N; g++) if ((s.mask >> g & 1) && ends(d, g, s.c1, s.c2, &lo[g], &hi[g])) got |= 1u << g;
        if (__builtin_popcount(got) < 2) continue;
        B.seg_iv.push_back(d.iv); B.seg_col.push_back(s.c1); B.seg_len.push_back(s.c2 - s.c1 + 1); B.seg_mask.push_back(got);
        for (int g = 0; g < N; g++) { const bool in = got >> g & 1; B.seg_left.push_back(in ? lo[g] : 0); B.seg_right.push_back(in ? hi[g] : 0); }
    }
    // canonical segment order: interval, first column, genome set
    {
        std::vector<size_t> idx(B.seg_iv.size());
        for (size_t i = 0; i < idx.size(); i++) idx[i] = i;
        auto before = [&](size_t x, size_t y) {
            if (B.seg_iv[x] != B.seg_iv[y]) return B.seg_iv[x] < B.seg_iv[y];
            if (B.seg_col[x] != B.seg_col[y]) return B.seg_col[x] < B.seg_col[y];
            return B.seg_mask[x] < B.seg_mask[y];
        };
        if (!std::is_sorted(idx.begin(), idx.end(), before)) {             // (the sweep hands them out nearly in this order already)
            std::sort(idx.begin(), idx.end(), before);
            mauve_ctx::BackboneResult S; S.N = N;
            for (size_t i : idx) {
                S.seg_iv.push_back(B.seg_iv[i]); S.seg_col.push_back(B.seg_col[i]); S.seg_len.push_back(B.seg_len[i]); S.seg_mask.push_back(B.seg_mask[i]);
                S.seg_left.insert(S.seg_left.end(), B.seg_left.begin() + (std::ptrdiff_t)(i * N), B.seg_left.begin() + (std::ptrdiff_t)((i + 1) * N));
                S.seg_right.insert(S.seg_right.end(), B.seg_right.begin() + (std::ptrdiff_t)(i * N), B.seg_right.begin() + (std::ptrdiff_t)((i + 1) * N));
            }
            B = std::move(S);
        }
    }
    B.islands.reserve(isl.size() * 8);
    for (size_t k = 0; k < isl.size(); k++) {
        const BbRec &q = isl[k];
        const BbIv &d = ivs[isl_ivx[k]];
        const int who = (int)(q.packed >> 16 & 0xff);
        int64_t lo = 0, hi = 0;
        kb = isl_ivx[k]; ka = q_isl + 2 * k; kz = ka + 1;
        (void)ends(d, who, q.c_first, q.c_last, &lo, &hi);
        const int64_t row[8] = {(int64_t)q.iv, (int64_t)(q.packed & 0xff), (int64_t)(q.packed >> 8 & 0xff), who, (int64_t)q.c_first, (int64_t)q.c_last, lo, hi};
        B.islands.insert(B.islands.end(), row, row + 8);
    }
    B.valid = true;
    if (trace) fprintf(stderr, "[trace] backbone: coordinates + tables %.3f ms\n", now_ms() - tb3);
    return MAUVE_OK;
}

// A caller's alignment (the _alignment entry points; the mirror hands over iv.Columns() of user Intervals): every genome's residue count
// in an interval's columns must be what its ends say -- right - left + 1, none for an absent genome -- and no column may carry a bit at or
// above nseq.  The homology pass turns these counts into base addresses (hom_column), so an inconsistent array would read outside the genomes.
static int check_columns(mauve_ctx *c, const char *who, int N, int64_t n_iv, const int64_t *left, const int64_t *right, const int64_t *col_off, const uint32_t *d_cols)
{
    const int64_t n_cols = col_off[n_iv];
    auto up = [](size_t x) { return (x + 63) & ~(size_t)63; };
    const size_t o_cnt = up(((size_t)n_iv + 1) * 8), o_bad = o_cnt + up((size_t)n_iv * N * 8), total = o_bad + 64;
    HIPCHK(c, c->bb_work.ensure(total));
    char *wk = c->bb_work.as<char>();
    HIPCHK(c, hipMemcpyAsync(wk, col_off, ((size_t)n_iv + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(wk + o_cnt, 0, total - o_cnt, c->stream));
    if (n_cols) {
        hipLaunchKernelGGL(bb_iv_residues, dim3((uint32_t)((n_cols + BB_CHUNK - 1) / BB_CHUNK)), dim3(256), 0, c->stream, d_cols, n_cols, N, reinterpret_cast<const int64_t *>(wk), n_iv,
                           reinterpret_cast<unsigned long long *>(wk + o_cnt), reinterpret_cast<uint32_t *>(wk + o_bad));
        HIPCHK(c, hipGetLastError());
    }
    std::vector<unsigned long long> cnt((size_t)n_iv * N); uint32_t bad = 0;
    HIPCHK(c, hipMemcpyAsync(cnt.data(), wk + o_cnt, cnt.size() * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&bad, wk + o_bad, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (bad) { c->err = std::string(who) + ": a column carries a genome bit at or above nseq"; return MAUVE_ERR_ARG; }
    for (int64_t iv = 0; iv < n_iv; iv++) for (int g = 0; g < N; g++) {
        const int64_t l = left[(size_t)(iv * N + g)], r = right[(size_t)(iv * N + g)];
        const int64_t want = l ? r - l + 1 : 0;
        if (l < 0 || want < 0 || (int64_t)cnt[(size_t)(iv * N + g)] != want) {
            c->err = std::string(who) + ": the columns of interval " + std::to_string(iv) + " hold " + std::to_string(cnt[(size_t)(iv * N + g)]) + " residues of genome " + std::to_string(g) +
                     ", its ends say " + std::to_string(want);
            return MAUVE_ERR_ARG;
        }
    }
    return MAUVE_OK;
}

extern "C" {

int mauve_backbone(mauve_ctx *c, int64_t island_gap_size, int64_t *n_seg, int64_t *n_islands)
{
    if (!c || !n_seg || !n_islands) return MAUVE_ERR_ARG;
    if (island_gap_size < 0 || island_gap_size > 0x7fffffff) { c->err = "backbone: island_gap_size out of range"; return MAUVE_ERR_ARG; }
    AlignResult &R = c->res;
    if (R.stale) { c->err = "backbone: the genomes were replaced after this alignment was made"; return MAUVE_ERR_STATE; }
    const int64_t n_iv = R.sz.n_iv;
    if ((int64_t)R.col_off.size() != n_iv + 1) { c->err = "backbone: no alignment in this context"; return MAUVE_ERR_STATE; }
    if (n_iv == 0) { c->bb = mauve_ctx::BackboneResult(); c->bb.N = c->nseq; c->bb.valid = true; *n_seg = *n_islands = 0; return MAUVE_OK; }   // nothing was aligned
    const int N = (int)(R.iv_left.size() / (size_t)n_iv);
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t *d_cols;
    if (R.cols_pending) d_cols = c->res_cols.as<uint32_t>();                // still where the assembly stage wrote them
    else {
        const size_t nb = (size_t)R.col_off[(size_t)n_iv] * 4;
        HIPCHK(c, c->bb_cols.ensure(nb + 64));
        if (nb) HIPCHK(c, hipMemcpyAsync(c->bb_cols.p, R.cols_data(), nb, hipMemcpyHostToDevice, c->stream));
        d_cols = c->bb_cols.as<uint32_t>();
    }
    const int rc = backbone_run(c, N, n_iv, R.iv_left.data(), R.iv_right.data(), R.iv_reverse.data(), R.col_off.data(), d_cols, island_gap_size);
    if (rc) return rc;
    *n_seg = (int64_t)c->bb.seg_iv.size(); *n_islands = (int64_t)c->bb.islands.size() / 8;
    return MAUVE_OK;
}

int mauve_backbone_alignment(mauve_ctx *c, int nseq, int64_t n_iv, const int64_t *left, const int64_t *right, const int8_t *reverse,
                             const int64_t *col_off, const uint32_t *cols, int64_t island_gap_size, int64_t *n_seg, int64_t *n_islands)
{
    if (!c || !n_seg || !n_islands || nseq < 1 || nseq > MAUVE_MAX_SEQ || n_iv < 0 || (n_iv && (!left || !right || !reverse || !col_off || !cols))) return MAUVE_ERR_ARG;
    c->bb = mauve_ctx::BackboneResult(); c->bb.N = nseq;
    *n_seg = *n_islands = 0;
    if (n_iv == 0) { c->bb.valid = true; return MAUVE_OK; }
    for (int64_t iv = 0; iv < n_iv; iv++) if (col_off[iv + 1] < col_off[iv] || col_off[0] != 0) { c->err = "backbone: col_off must ascend from 0"; return MAUVE_ERR_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nb = (size_t)col_off[n_iv] * 4;
    HIPCHK(c, c->bb_cols.ensure(nb + 64));
    if (nb) HIPCHK(c, hipMemcpyAsync(c->bb_cols.p, cols, nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));                            // cols is the caller's (pageable) memory
    if (const int rcc = check_columns(c, "backbone", nseq, n_iv, left, right, col_off, c->bb_cols.as<uint32_t>())) return rcc;
    const int rc = backbone_run(c, nseq, n_iv, left, right, reverse, col_off, c->bb_cols.as<uint32_t>(), island_gap_size);
    if (rc) return rc;
    *n_seg = (int64_t)c->bb.seg_iv.size(); *n_islands = (int64_t)c->bb.islands.size() / 8;
    return MAUVE_OK;
}


void mauve_hmm_params_from(double identity, double pgh, double pgu, mauve_hmm_params *h)
{
    if (!h) return;
    if (!(identity > 0.25 && identity < 1.0) || !(pgh > 0.0 && pgh <= 1.0) || !(pgu > 0.0 && pgu <= 1.0)) {
        h->match = h->mismatch = h->gap = 0; h->go_homologous = h->go_unrelated = 1;
        return;
    }
    h->match = (int32_t)lround(1000.0 * log(identity / 0.25));
    h->mismatch = (int32_t)lround(1000.0 * log((1.0 - identity) / 0.75));
    h->gap = -500;
    h->go_homologous = (int32_t)lround(1000.0 * log(pgh));
    h->go_unrelated = (int32_t)lround(1000.0 * log(pgu));
}

// the work behind mauve_apply_homology / mauve_apply_homology_alignment: columns on the device (n_cols entries), interval table on the
// host; the re-split columns are left in c->hom_cols (*n_new of them, offsets in noff) unless nothing moved
static int homology_core(mauve_ctx *c, int N, int64_t n_iv, const int64_t *left, const int64_t *right, const int8_t *reverse, const int64_t *col_off,
                         const uint32_t *d_cols, const mauve_hmm_params *h, std::vector<int64_t> &noff, int64_t *n_new_out, int64_t *moved_out)
{
    const int64_t n_cols = col_off[n_iv];
    *moved_out = 0; *n_new_out = n_cols;
    if (h->go_homologous > 0 || h->go_unrelated > 0) { c->err = "apply_homology: transition scores are log probabilities (<= 0)"; return MAUVE_ERR_ARG; }
    if (N != c->nseq) { c->err = "apply_homology: the alignment does not belong to the genomes of this context"; return MAUVE_ERR_STATE; }
    if (n_cols == 0) return MAUVE_OK;
    // work items: (interval with >= 2 genomes, pair, segment of 4096 columns)
    if (std::abs((int64_t)h->match) > 65536 || std::abs((int64_t)h->mismatch) > 65536 || std::abs((int64_t)h->gap) > 65536 || h->go_homologous < -(1 << 28) || h->go_unrelated < -(1 << 28)) {
        c->err = "apply_homology: scores beyond +-65536 (transitions beyond -2^28) are not supported"; return MAUVE_ERR_ARG;
    }
    std::vector<HomIv> ivs; std::vector<HomItem> items; int64_t residues = 0; uint32_t n_segs = 0;
    struct PairRun { uint32_t first, nseg; };                  // the items of one (interval, pair): consecutive, segment order
    std::vector<PairRun> runs;
    for (int64_t iv = 0; iv < n_iv; iv++) {
        int g[MAUVE_MAX_SEQ], n = 0;
        for (int x = 0; x < N; x++) if (left[(size_t)(iv * N + x)]) {
            const int64_t l = left[(size_t)(iv * N + x)], r = right[(size_t)(iv * N + x)];
            if (l < 1 || r < l || r > c->lens[(size_t)x]) { c->err = "apply_homology: an interval lies outside its genome"; return MAUVE_ERR_ARG; }
            g[n++] = x; residues += r - l + 1;
        }
        const int64_t nc = col_off[(size_t)iv + 1] - col_off[(size_t)iv];
        if (n < 2 || nc <= 0) continue;
        const int64_t nsg = (nc + HOM_SEG - 1) / HOM_SEG;
        if ((int64_t)n_segs + nsg > 0x7fffffff || (int64_t)items.size() + nsg * n * (n - 1) / 2 > 0x7fffffff) { c->err = "apply_homology: too many segments"; return MAUVE_ERR_LIMIT; }
        ivs.push_back(HomIv{col_off[(size_t)iv], nc, (uint32_t)iv, (uint32_t)nsg, n_segs, 0u});
        n_segs += (uint32_t)nsg;
        for (int x = 0; x < n; x++) for (int y = x + 1; y < n; y++) {
            runs.push_back(PairRun{(uint32_t)items.size(), (uint32_t)nsg});
            for (int64_t sgi = 0; sgi < nsg; sgi++) items.push_back(HomItem{(uint32_t)(ivs.size() - 1), (uint32_t)sgi, (uint8_t)g[x], (uint8_t)g[y], 0});
        }
    }
    const size_t n_items = items.size();
    const size_t n_tiles = (size_t)((n_cols + BB_CHUNK - 1) / BB_CHUNK);
    auto up = [](size_t x) { return (x + 63) & ~(size_t)63; };
    const size_t o_items = up(ivs.size() * sizeof(HomIv)), o_left = o_items + up(n_items * sizeof(HomItem)), o_right = o_left + up((size_t)n_iv * N * 8),
                 o_rev = o_right + up((size_t)n_iv * N * 8), o_off = o_rev + up((size_t)n_iv * N), o_noff = o_off + up(((size_t)n_iv + 1) * 8), o_tile = o_noff + up(((size_t)n_iv + 1) * 8),
                 o_base = o_tile + up(n_tiles * 4), o_cnt = o_base + up((n_tiles + 1) * 8), o_keep = o_cnt + 64, o_scnt = o_keep + up((size_t)n_cols * 4),
                 o_rank = o_scnt + up((size_t)n_segs * N * 4), o_fn = o_rank + up((size_t)n_segs * N * 8), o_xin = o_fn + up(n_items * sizeof(HomFn)), o_smap = o_xin + up(n_items * 4),
                 o_send = o_smap + up(n_items), o_step = o_send + up(n_items), o_pred = o_step + up(n_items * 64), total = o_pred + n_items * 64 * 16 + 64;
    HIPCHK(c, c->bb_work.ensure(total));
    char *wk = c->bb_work.as<char>();
    HIPCHK(c, hipMemcpyAsync(wk, ivs.data(), ivs.size() * sizeof(HomIv), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(wk + o_items, items.data(), n_items * sizeof(HomItem), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(wk + o_left, left, (size_t)n_iv * N * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(wk + o_right, right, (size_t)n_iv * N * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(wk + o_rev, reverse, (size_t)n_iv * N, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(wk + o_off, col_off, ((size_t)n_iv + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(wk + o_cnt, 0, 64, c->stream));
    HIPCHK(c, hipMemsetAsync(wk + o_keep, 0, (size_t)n_cols * 4, c->stream));
    uint32_t *keep = reinterpret_cast<uint32_t *>(wk + o_keep);
    if (n_items) {
        const HomIv *d_ivs = reinterpret_cast<const HomIv *>(wk); const HomItem *d_items = reinterpret_cast<const HomItem *>(wk + o_items);
        const int64_t *d_left = reinterpret_cast<const int64_t *>(wk + o_left), *d_right = reinterpret_cast<const int64_t *>(wk + o_right);
        const int8_t *d_rev = reinterpret_cast<const int8_t *>(wk + o_rev);
        HomGenomes gw; memset(&gw, 0, sizeof gw);
        for (int g = 0; g < c->nseq; g++) gw.word_off[g] = c->word_off[(size_t)g];
        const HomScores sc{h->match, h->mismatch, h->gap, h->go_homologous, h->go_unrelated};
        // residues per segment -> base positions at the segment starts
        hipLaunchKernelGGL(hom_seg_count, dim3(n_segs), dim3(64), 0, c->stream, d_cols, d_ivs, (uint32_t)ivs.size(), N, reinterpret_cast<uint32_t *>(wk + o_scnt));
        HIPCHK(c, hipGetLastError());
        std::vector<uint32_t> scnt((size_t)n_segs * N); std::vector<int64_t> rank0((size_t)n_segs * N);
        HIPCHK(c, hipMemcpyAsync(scnt.data(), wk + o_scnt, scnt.size() * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (const HomIv &iv : ivs)
            for (int g = 0; g < N; g++) { int64_t run = 0; for (uint32_t sgi = 0; sgi < iv.nseg; sgi++) { rank0[(size_t)(iv.seg0 + sgi) * N + g] = run; run += scnt[(size_t)(iv.seg0 + sgi) * N + g]; } }
        HIPCHK(c, hipMemcpyAsync(wk + o_rank, rank0.data(), rank0.size() * 8, hipMemcpyHostToDevice, c->stream));
        const uint32_t grid = (uint32_t)std::min<size_t>(n_items, 256 * 64);
        const int64_t *d_rank = reinterpret_cast<const int64_t *>(wk + o_rank);
        hipLaunchKernelGGL(hom_fn, dim3(grid), dim3(64), 0, c->stream, d_cols, d_items, (uint32_t)n_items, d_ivs, d_left, d_right, d_rev, N, c->genomes.as<uint64_t>(), gw, sc, d_rank,
                           reinterpret_cast<HomFn *>(wk + o_fn));
        HIPCHK(c, hipGetLastError());
        std::vector<HomFn> fn(n_items); std::vector<int32_t> xin(n_items); std::vector<uint8_t> fin(runs.size());
        HIPCHK(c, hipMemcpyAsync(fn.data(), wk + o_fn, n_items * sizeof(HomFn), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (size_t r = 0; r < runs.size(); r++) {              // x at every segment start; the path starts in U: x below go_homologous
            int64_t x = (int64_t)h->go_homologous - 1;
            for (uint32_t k = 0; k < runs[r].nseg; k++) {
                const HomFn &f = fn[runs[r].first + k];
                xin[runs[r].first + k] = (int32_t)x;
                x = std::min<int64_t>(std::max<int64_t>(x + f.t, f.lo), f.hi);
            }
            fin[r] = x >= 0;
        }
        HIPCHK(c, hipMemcpyAsync(wk + o_xin, xin.data(), n_items * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(hom_pred, dim3(grid), dim3(64), 0, c->stream, d_cols, d_items, (uint32_t)n_items, d_ivs, d_left, d_right, d_rev, N, c->genomes.as<uint64_t>(), gw, sc, d_rank,
                           reinterpret_cast<const int32_t *>(wk + o_xin), reinterpret_cast<uint64_t *>(wk + o_pred), reinterpret_cast<uint8_t *>(wk + o_step),
                           reinterpret_cast<uint8_t *>(wk + o_smap));
        HIPCHK(c, hipGetLastError());
        std::vector<uint8_t> smap(n_items), send(n_items);
        HIPCHK(c, hipMemcpyAsync(smap.data(), wk + o_smap, n_items, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (size_t r = 0; r < runs.size(); r++) {              // state at every segment's last column, from the interval's end backwards
            uint32_t e = fin[r];
            for (uint32_t k = runs[r].nseg; k-- > 0;) { send[runs[r].first + k] = (uint8_t)e; e = (smap[runs[r].first + k] >> e) & 1u; }
        }
        HIPCHK(c, hipMemcpyAsync(wk + o_send, send.data(), n_items, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(hom_keep, dim3(grid), dim3(64), 0, c->stream, d_cols, d_items, (uint32_t)n_items, d_ivs, reinterpret_cast<const uint64_t *>(wk + o_pred),
                           reinterpret_cast<const uint8_t *>(wk + o_step), reinterpret_cast<const uint8_t *>(wk + o_send), keep);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));             // (send / xin are host vectors)
    }
    hipLaunchKernelGGL(hom_count, dim3((uint32_t)n_tiles), dim3(256), 0, c->stream, d_cols, keep, n_cols, reinterpret_cast<uint32_t *>(wk + o_tile),
                       reinterpret_cast<unsigned long long *>(wk + o_cnt));
    HIPCHK(c, hipGetLastError());
    std::vector<uint32_t> tile_out(n_tiles); unsigned long long moved = 0;
    HIPCHK(c, hipMemcpyAsync(tile_out.data(), wk + o_tile, n_tiles * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&moved, wk + o_cnt, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *moved_out = (int64_t)moved;
    if (moved == 0) return MAUVE_OK;                          // every residue stays where it is
    std::vector<int64_t> tile_base(n_tiles + 1, 0);
    for (size_t t = 0; t < n_tiles; t++) tile_base[t + 1] = tile_base[t] + tile_out[t];
    const int64_t n_new = tile_base[n_tiles];
    if (n_new > residues) { c->err = "apply_homology: more columns than residues"; return MAUVE_ERR_STATE; }
    HIPCHK(c, c->hom_cols.ensure(((size_t)n_new + 64) * 4));
    HIPCHK(c, hipMemcpyAsync(wk + o_base, tile_base.data(), (n_tiles + 1) * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(hom_write, dim3((uint32_t)n_tiles), dim3(256), 0, c->stream, d_cols, keep, n_cols, reinterpret_cast<const int64_t *>(wk + o_base), c->hom_cols.as<uint32_t>());
    hipLaunchKernelGGL(hom_offsets, dim3((uint32_t)std::min<int64_t>(n_iv + 1, 4096)), dim3(64), 0, c->stream, d_cols, keep, n_cols, reinterpret_cast<const int64_t *>(wk + o_base),
                       reinterpret_cast<const int64_t *>(wk + o_off), n_iv + 1, reinterpret_cast<int64_t *>(wk + o_noff));
    HIPCHK(c, hipGetLastError());
    noff.resize((size_t)n_iv + 1);
    HIPCHK(c, hipMemcpyAsync(noff.data(), wk + o_noff, ((size_t)n_iv + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (noff[0] != 0 || noff[(size_t)n_iv] != n_new) { c->err = "apply_homology: column offsets do not add up"; return MAUVE_ERR_STATE; }
    *n_new_out = n_new;
    return MAUVE_OK;
}

int mauve_apply_homology(mauve_ctx *c, const mauve_hmm_params *h, mauve_align_sizes *sizes, int64_t *n_moved)
{
    if (!c || !h) return MAUVE_ERR_ARG;
    AlignResult &R = c->res;
    if (R.stale || R.genomes_replaced) { c->err = "apply_homology: the genomes were replaced after this alignment was made"; return MAUVE_ERR_STATE; }
    const int64_t n_iv = R.sz.n_iv;
    if ((int64_t)R.col_off.size() != n_iv + 1) { c->err = "apply_homology: no alignment in this context"; return MAUVE_ERR_STATE; }
    if (n_moved) *n_moved = 0;
    if (sizes) *sizes = R.sz;
    if (n_iv == 0) return MAUVE_OK;
    const int N = (int)(R.iv_left.size() / (size_t)n_iv);
    const int64_t n_cols = R.col_off[(size_t)n_iv];
    HIPCHK(c, hipSetDevice(c->device));
    if (!R.cols_pending && n_cols) {                          // the columns back to where the assembly stage keeps them
        HIPCHK(c, c->res_cols.ensure(((size_t)n_cols + 64) * 4));
        HIPCHK(c, hipMemcpyAsync(c->res_cols.p, R.cols_data(), (size_t)n_cols * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    std::vector<int64_t> noff; int64_t n_new = 0, moved = 0;
    const int rc = homology_core(c, N, n_iv, R.iv_left.data(), R.iv_right.data(), R.iv_reverse.data(), R.col_off.data(), c->res_cols.as<uint32_t>(), h, noff, &n_new, &moved);
    if (rc) return rc;
    if (n_moved) *n_moved = moved;
    if (moved == 0) return MAUVE_OK;
    // the new columns are the result's columns from here on, resident like a device-assembled result
    std::swap(c->res_cols, c->hom_cols);
    R.col_off.swap(noff);
    R.n_cols = (size_t)n_new; R.sz.n_cols = n_new;
    R.cols_pending = true; R.cols_ext = nullptr; R.cols_fill = 0; R.cols_dirty.clear();
    c->bb = mauve_ctx::BackboneResult();                       // a backbone of the old columns no longer applies
    if (sizes) *sizes = R.sz;
    return MAUVE_OK;
}

int mauve_apply_homology_alignment(mauve_ctx *c, int nseq, int64_t n_iv, const int64_t *left, const int64_t *right, const int8_t *reverse, const int64_t *col_off,
                                   const uint32_t *cols, const mauve_hmm_params *h, int64_t *col_off_out, uint32_t *cols_out, int64_t *n_moved)
{
    if (!c || !h || nseq < 1 || nseq > MAUVE_MAX_SEQ || n_iv < 0 || !col_off_out || (n_iv && (!left || !right || !reverse || !col_off || !cols || !cols_out))) return MAUVE_ERR_ARG;
    if (n_moved) *n_moved = 0;
    col_off_out[0] = 0;
    if (n_iv == 0) return MAUVE_OK;
    for (int64_t iv = 0; iv < n_iv; iv++) if (col_off[iv + 1] < col_off[iv] || col_off[0] != 0) { c->err = "apply_homology: col_off must ascend from 0"; return MAUVE_ERR_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    const int64_t n_cols = col_off[n_iv];
    HIPCHK(c, c->bb_cols.ensure((size_t)n_cols * 4 + 64));
    if (n_cols) HIPCHK(c, hipMemcpyAsync(c->bb_cols.p, cols, (size_t)n_cols * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));                            // cols is the caller's (pageable) memory
    if (nseq != c->nseq) { c->err = "apply_homology: the alignment does not belong to the genomes of this context"; return MAUVE_ERR_STATE; }
    if (const int rcc = check_columns(c, "apply_homology", nseq, n_iv, left, right, col_off, c->bb_cols.as<uint32_t>())) return rcc;
    std::vector<int64_t> noff; int64_t n_new = 0, moved = 0;
    const int rc = homology_core(c, nseq, n_iv, left, right, reverse, col_off, c->bb_cols.as<uint32_t>(), h, noff, &n_new, &moved);
    if (rc) return rc;
    if (n_moved) *n_moved = moved;
    if (moved == 0) { memcpy(col_off_out, col_off, ((size_t)n_iv + 1) * 8); if (n_cols) memcpy(cols_out, cols, (size_t)n_cols * 4); return MAUVE_OK; }
    memcpy(col_off_out, noff.data(), ((size_t)n_iv + 1) * 8);
    HIPCHK(c, hipMemcpy(cols_out, c->hom_cols.p, (size_t)n_new * 4, hipMemcpyDeviceToHost));
    return MAUVE_OK;
}

int mauve_backbone_fetch(mauve_ctx *c, int64_t *seg_iv, int64_t *seg_col, int64_t *seg_len, uint32_t *seg_mask, int64_t *seg_left,
                         int64_t *seg_right, int64_t *islands)
{
    if (!c) return MAUVE_ERR_ARG;
    const mauve_ctx::BackboneResult &B = c->bb;
    if (!B.valid) { c->err = "backbone_fetch: no backbone computed"; return MAUVE_ERR_STATE; }
    if (seg_iv) std::copy(B.seg_iv.begin(), B.seg_iv.end(), seg_iv);
    if (seg_col) std::copy(B.seg_col.begin(), B.seg_col.end(), seg_col);
    if (seg_len) std::copy(B.seg_len.begin(), B.seg_len.end(), seg_len);
    if (seg_mask) std::copy(B.seg_mask.begin(), B.seg_mask.end(), seg_mask);
    if (seg_left) std::copy(B.seg_left.begin(), B.seg_left.end(), seg_left);
    if (seg_right) std::copy(B.seg_right.begin(), B.seg_right.end(), seg_right);
    if (islands) std::copy(B.islands.begin(), B.islands.end(), islands);
    return MAUVE_OK;
}

}  // extern "C"
