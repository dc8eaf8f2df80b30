// dp_batch.hip -- batched gapped alignment of inter-anchor intervals on gfx950.
//
// Stands behind the mems::GappedAligner seam (Aligner::SetGappedAligner, mauveAligner.cpp:674;
// MuscleInterface::Align, MatchRecord.h:302-321; CallMuscleFast, repeatoire.cpp:1262): libMUSCLE is
// [EXT]; the frozen replacement is a progressive N-way alignment in genome order whose
// every step is a 3-state affine-gap (Gotoh) DP of the current profile against the next sequence,
// sum-of-pairs scores from per-column base counts, predecessor preference M > X > Y on ties.
//
// Mapping: the basic unit is one wave64 per interval (small intervals share a wave, dp_groups; the longest get a
// 16-wave workgroup, dp_step_big -- same recurrence, different schedules).  The profile columns are the DP rows; the 64 lanes hold 64
// consecutive rows and sweep the sequence as a systolic anti-diagonal wavefront: lane l works on
// column j = t - l at step t, takes (i-1, j) from lane l-1 by a wave shuffle, (i-1, j-1) from what it
// took one step earlier and (i, j-1) from its own registers.  Profiles longer than 64 columns run in
// stripes; the last row of a stripe is parked in a per-interval HBM row buffer.  Traceback bytes are
// written anti-diagonal-major (one coalesced 64-byte store per step), walked back by the wave, and
// the new profile is rebuilt in forward order with ballot prefix counts.  Integer VALU + shuffle
// bound, not HBM and not MFMA.
#include "common.hpp"
#include "dev_scan.hpp"
#include <algorithm>
#include <functional>
#include <cstring>
#include <cstdlib>

#define DP_NEG_INF (-(1 << 29))
constexpr int DP_LDS_TB = 8192;             // per wave: 128 systolic steps x 64 lanes.  With 12288 the LDS held dp_step to 3
                                            // workgroups per CU; 8192 (and <= 128 VGPRs) gives 4 -- C3's DP stage 3.9 -> 2.4 ms
constexpr int DP_LDS_OPS = 512;            // 4 groups x 128 reversed ops (dp_groups); >= 128 for the one-wave path

struct DpMeta {
    int32_t m;        // current profile length
    int32_t krows;    // sequences merged so far
    int32_t cur;      // which profile buffer is current (0 = A, 1 = B)
    int32_t pad;
    int64_t score;
    int64_t cells;
};

struct DpScoring { int32_t go, ge; int32_t s[4][4]; };

// ---- banded steps: intervals whose longest sequence exceeds max_gapped_len run every progressive step
// inside the band |j - c(i)| <= W around the scaled diagonal c(i) = floor(i * n / m), W = 128 + |n - m| + (m + n) / 64:
// cells outside are minus infinity.  Only the band is swept and only its traceback is stored: stripe s covers the
// columns J0(s) .. J1(s), its traceback stride is dp_stride(m, n, banded) steps (uniform over the stripes).
__host__ __device__ __forceinline__ int64_t dp_band_w(int64_t m, int64_t n) { return 128 + (m > n ? m - n : n - m) + (m + n) / 64; }
__host__ __device__ __forceinline__ int64_t dp_stride(int64_t m, int64_t n, bool banded)
{
    const int64_t full = n + 64;
    if (!banded || m < 1) return full;
    const int64_t b = (63 * n) / m + 2 * dp_band_w(m, n) + 72;
    return b < full ? b : full;
}
__host__ __device__ __forceinline__ int64_t dp_diag(int64_t i, int64_t m, int64_t n) { return (i * n) / m; }
// first column of stripe s's sweep: one left of its first row's band, so that the diagonal input of the band's first
// cell (row above, one column left -- in that row's band whenever c() steps there) has been seen
__host__ __device__ __forceinline__ int32_t dp_j0(int32_t s, int64_t m, int64_t n, bool banded)
{
    if (!banded) return 0;
    const int64_t j = dp_diag((int64_t)s * 64 + 1, m, n) - dp_band_w(m, n) - 1;
    return (int32_t)(j > 0 ? j : 0);
}
// traceback bytes an interval step may need when only bounds of the profile length are known (mmin <= m <= mmax)
__host__ __device__ __forceinline__ int64_t dp_tb_need(int64_t mmin, int64_t mmax, int64_t n, bool banded)
{
    const int64_t stripes = (mmax + 63) / 64;
    int64_t stride = n + 64;
    if (banded && mmin >= 1) {
        const int64_t d0 = n > mmin ? n - mmin : mmin - n, d1 = n > mmax ? n - mmax : mmax - n;
        const int64_t wmax = 128 + (d0 > d1 ? d0 : d1) + (mmax + n) / 64;
        const int64_t b = (63 * n) / mmin + 2 * wmax + 72;
        if (b < stride) stride = b;
    }
    return stripes * stride * 64;
}

// cells a step evaluates: m x n, or the band's share of it (rows 1..m, columns max(1, blo)..bhi); every lane returns the total
__device__ __forceinline__ int64_t dp_step_cells(int32_t m, int32_t n, bool banded, int lane)
{
    if (!banded) return (int64_t)m * n;
    const int64_t W = dp_band_w(m, n);
    int64_t acc = 0;
    for (int64_t i = 1 + lane; i <= m; i += 64) {
        const int64_t c = dp_diag(i, m, n);
        const int64_t lo = c - W > 1 ? c - W : 1, hi = c + W < n ? c + W : n;
        if (hi >= lo) acc += hi - lo + 1;
    }
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    return acc;
}

// Kernel class of an interval from an ESTIMATE of its profile lengths: aligning a profile with one more sequence rarely
// makes it much longer than the longest sequence so far, while the safe bound is the SUM of the lengths -- which puts
// nearly every 5-way interval of ~20-base gaps into the one-wave class.  est(m) = min(bound, longest + longest / 8 + 2);
// the kernels check the real lengths and fall back (dp_groups).  3: four per wave (rows <= 16), 2: two per wave
// (rows <= 32), 1: a wave of its own.
struct DpClassEst {
    int64_t mbound = 0, longest = 0, rows = 0, steps = 0, nmax = 0; bool first = true;
    int mode = 0;                     // 0: the estimate; 1: the safe bound (MAUVE_DP_CLASS=bound); 2: half the longest (=wild: tests the fallback)
    __host__ __device__ void add(int64_t n)
    {
        if (n == 0) return;
        if (first) { first = false; mbound = longest = n; return; }
        int64_t e = mode == 1 ? mbound : (mode == 2 ? longest / 2 + 1 : longest + longest / 8 + 2);
        if (e > mbound) e = mbound;
        if (e > rows) rows = e;
        if (e + n > steps) steps = e + n;
        if (n > nmax) nmax = n;
        mbound += n; if (n > longest) longest = n;
    }
    // systolic kernels (MAUVE_DP_OLD): 4: four per wave (rows <= 16), 3: two per wave (rows <= 32), 1: a wave of its own
    __host__ __device__ int klass(int tmax) const { return steps <= tmax ? (rows <= 16 ? 4 : (rows <= 32 ? 3 : 1)) : 1; }
    // register-blocked kernels: G lanes x 4 rows per interval, n + G steps in the LDS slice.  4: G = 4 (sixteen per wave),
    // 3: G = 8, 2: G = 16, 1: a wave of its own
    __host__ __device__ int klass2(int rows_per_lane, int tcap) const
    {
        if (rows <= 4 * rows_per_lane && nmax + 4 <= tcap) return 4;
        if (rows <= 8 * rows_per_lane && nmax + 8 <= tcap) return 3;
        if (rows <= 16 * rows_per_lane && nmax + 16 <= tcap) return 2;
        return 1;
    }
};

// lane l receives lane l-1's value; wave_shr1z: lane 0 receives 0 (its caller puts the real input there), wave_shr1: lane 0
// keeps its own (gfx9 DPP wave_shr:1)
__device__ __forceinline__ int32_t wave_shr1z(int32_t v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ int32_t wave_shr1(int32_t v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }

// v with lane 0 replaced by the wave-uniform value x (v_writelane_b32; this compiler has no builtin for it)
__device__ __forceinline__ int32_t lane0_set(int32_t v, int32_t x)
{
    x = __builtin_amdgcn_readfirstlane(x);               // (wave-uniform by contract; this pins it to a scalar register)
    asm("" : "+s"(x));                                   // a compile-time constant would be folded into the operand: v_writelane takes no literal
    asm("v_writelane_b32 %0, %1, 0" : "+v"(v) : "s"(x));
    return v;
}

__device__ __forceinline__ void max3(int32_t a, int32_t b, int32_t c, int32_t &best, uint32_t &p)
{
    best = a; p = 0;
    if (b > best) { best = b; p = 1; }
    if (c > best) { best = c; p = 2; }
}

// One 64-row stripe of a DP step as a wave sees it: the lane's row constants, the rolling cell state and the two
// 64-column chunks (current, next) of what lane 0 consumes -- the sequence bases and the boundary row above the
// stripe.  Chunks are fetched by the whole wave one round (64 steps) ahead, from clamped addresses so the load is
// unconditional and its wait lands a round later, and handed to lane 0 with v_readlane.
struct DpStripe {
    int32_t s, i, m, n, steps, gyo, gye, gxo, gxe, sub0, sub1, sub2, sub3;
    int32_t j0, blo, bhi, plo, phi;          // first column of the stripe's sweep; this row's band; the band of the row above the stripe
    bool active, park;
    const uint8_t *seq; const int32_t *rin; int32_t *rout; uint8_t *tbs;
    int32_t Mc, Xc, Yc, Md, Xd, Yd;
    uint32_t bcur, sq_cur, sq_nxt;
    int32_t bM_cur, bX_cur, bY_cur, bM_nxt, bX_nxt, bY_nxt;
};

template <bool BANDED>
__device__ __forceinline__ void dp_stripe_begin(DpStripe &S, int32_t s, int lane, int32_t m, int32_t n, int32_t nstripes,
                                                const uint32_t *Pc, const uint8_t *seq, const DpScoring &sc, int32_t krows,
                                                int32_t *rowbuf, uint8_t *tbp, int32_t T)
{
    S.s = s; S.m = m; S.n = n; S.seq = seq;
    S.i = s * 64 + lane + 1;
    S.active = S.i <= m;
    const uint32_t cn = S.active ? Pc[S.i - 1] : 0u;
    const int32_t c0 = cn & 255, c1 = (cn >> 8) & 255, c2 = (cn >> 16) & 255, c3 = cn >> 24;
    const int32_t r = c0 + c1 + c2 + c3;
    // sum-of-pairs substitution score of this profile column against each base (named registers: a
    // runtime-indexed array would go to scratch)
    S.sub0 = c0 * sc.s[0][0] + c1 * sc.s[1][0] + c2 * sc.s[2][0] + c3 * sc.s[3][0];
    S.sub1 = c0 * sc.s[0][1] + c1 * sc.s[1][1] + c2 * sc.s[2][1] + c3 * sc.s[3][1];
    S.sub2 = c0 * sc.s[0][2] + c1 * sc.s[1][2] + c2 * sc.s[2][2] + c3 * sc.s[3][2];
    S.sub3 = c0 * sc.s[0][3] + c1 * sc.s[1][3] + c2 * sc.s[2][3] + c3 * sc.s[3][3];
    S.gxo = sc.go * r; S.gxe = sc.ge * r;
    S.gyo = sc.go * krows; S.gye = sc.ge * krows;
    S.rin = rowbuf + (size_t)((s & 1) ^ 1) * 3 * (n + 1);      // written by stripe s-1
    S.rout = rowbuf + (size_t)(s & 1) * 3 * (n + 1);
    S.park = s + 1 < nstripes;
    const int32_t rows_here = min(64, m - s * 64);
    if (BANDED) {
        const int64_t W = dp_band_w(m, n);
        const int64_t ci = dp_diag(S.active ? S.i : m, m, n);
        S.blo = (int32_t)max((int64_t)0, ci - W); S.bhi = (int32_t)min((int64_t)n, ci + W);
        const int64_t cp = dp_diag((int64_t)s * 64, m, n);
        S.plo = (int32_t)max((int64_t)0, cp - W); S.phi = (int32_t)min((int64_t)n, cp + W);
        S.j0 = dp_j0(s, m, n, true);
        const int32_t j1 = (int32_t)min((int64_t)n, dp_diag((int64_t)s * 64 + rows_here, m, n) + W);
        S.steps = (j1 - S.j0 + 1) + rows_here - 1;
    } else {
        S.blo = 0; S.bhi = n; S.plo = 0; S.phi = n; S.j0 = 0;
        S.steps = n + rows_here;                                // t = 0 .. n + rows_here - 1
    }
    S.tbs = tbp + (size_t)s * T * 64 + lane;
    S.Mc = S.Xc = S.Yc = S.Md = S.Xd = S.Yd = DP_NEG_INF;
    S.bcur = 0;
}

// chunk k of lane 0's inputs: base t-1 and boundary column t for t = 64k + lane
template <bool BANDED>
__device__ __forceinline__ void dp_stripe_chunk(const DpStripe &S, int32_t k, int lane, uint32_t &sq, int32_t &bM, int32_t &bX, int32_t &bY)
{
    const int32_t col = (BANDED ? S.j0 : 0) + 64 * k + lane;
    sq = (uint32_t)S.seq[min(max(col - 1, 0), S.n - 1)];
    const bool inr = !BANDED || (col >= S.plo && col <= S.phi);               // inside the band of the row above (always, unbanded)
    if (S.s == 0) {
        bM = col == 0 ? 0 : DP_NEG_INF; bX = DP_NEG_INF;
        bY = (col == 0 || !inr) ? DP_NEG_INF : S.gyo + (col - 1) * S.gye;
    } else {
        const int32_t cc = min(col, S.n);
        bM = inr ? S.rin[cc] : DP_NEG_INF; bX = inr ? S.rin[(S.n + 1) + cc] : DP_NEG_INF; bY = inr ? S.rin[2 * (S.n + 1) + cc] : DP_NEG_INF;
    }
}

// round c: steps t = 64c .. min(64c + 63, steps - 1).  The cell update is straight-line code: the state moves only where
// the lane has a cell (and, banded, the cell is inside the band), the traceback byte is stored unconditionally -- the
// slot (t, lane) belongs to this lane and this step alone and is only ever read for real cells -- and the value at
// (m, n) is simply what the lane of row m holds when the sweep is over.  Lane 0's inputs arrive by v_readlane from the
// chunk registers and are put in place by v_writelane; the loop counter and bounds are wave-uniform (scalar branch).
template <bool BANDED>
__device__ __forceinline__ void dp_stripe_round(DpStripe &S, int32_t c, int lane)
{
    if (c == 0) dp_stripe_chunk<BANDED>(S, 0, lane, S.sq_cur, S.bM_cur, S.bX_cur, S.bY_cur);
    else { S.sq_cur = S.sq_nxt; S.bM_cur = S.bM_nxt; S.bX_cur = S.bX_nxt; S.bY_cur = S.bY_nxt; }
    dp_stripe_chunk<BANDED>(S, c + 1, lane, S.sq_nxt, S.bM_nxt, S.bX_nxt, S.bY_nxt);
    const int32_t t0 = __builtin_amdgcn_readfirstlane(64 * c), t_end = __builtin_amdgcn_readfirstlane(min(64 * c + 64, S.steps));
    const bool park = __builtin_amdgcn_readfirstlane((int)S.park) != 0;
    const int32_t jbase = (BANDED ? S.j0 : 0) - lane;
    uint8_t *tbw = S.tbs + (size_t)t0 * 64;
    for (int32_t t = t0; t < t_end; t++, tbw += 64) {
        const int32_t j = jbase + t;
        // (i-1, j): lane-1's newest values (DPP wave shift); lane 0 takes the stripe's upper boundary row
        int32_t Mu = wave_shr1z(S.Mc), Xu = wave_shr1z(S.Xc), Yu = wave_shr1z(S.Yc);
        int32_t bn = wave_shr1z((int32_t)S.bcur);
        const int sel = t & 63;
        Mu = lane0_set(Mu, __builtin_amdgcn_readlane(S.bM_cur, sel));
        Xu = lane0_set(Xu, __builtin_amdgcn_readlane(S.bX_cur, sel));
        Yu = lane0_set(Yu, __builtin_amdgcn_readlane(S.bY_cur, sel));
        bn = lane0_set(bn, __builtin_amdgcn_readlane((int32_t)S.sq_cur, sel));
        const uint32_t bnext = (uint32_t)bn;
        S.bcur = bnext;
        const bool on = S.active && (uint32_t)j <= (uint32_t)S.n, j1 = j >= 1;
        int32_t best; uint32_t pm, px, py;
        max3(S.Md, S.Xd, S.Yd, best, pm);
        const int32_t sa = (bnext & 1) ? S.sub1 : S.sub0, sb = (bnext & 1) ? S.sub3 : S.sub2;
        int32_t Mn = max(best + ((bnext & 2) ? sb : sa), DP_NEG_INF);
        max3(Mu + S.gxo, Xu + S.gxe, Yu + S.gxo, best, px);
        const int32_t Xn = max(best, DP_NEG_INF);
        max3(S.Mc + S.gyo, S.Xc + S.gyo, S.Yc + S.gye, best, py);    // (i, j-1): own previous column
        int32_t Yn = max(best, DP_NEG_INF);
        Mn = j1 ? Mn : DP_NEG_INF; Yn = j1 ? Yn : DP_NEG_INF; pm = j1 ? pm : 0u; py = j1 ? py : 0u;
        *tbw = (uint8_t)(pm | (px << 2) | (py << 4));
        const bool upd = on && (!BANDED || (j >= S.blo && j <= S.bhi));
        if (park) { if (upd && lane == 63) { S.rout[j] = Mn; S.rout[(S.n + 1) + j] = Xn; S.rout[2 * (S.n + 1) + j] = Yn; } }
        if (BANDED) {                                                  // outside the band: minus infinity for whoever reads it
            S.Mc = upd ? Mn : (on ? DP_NEG_INF : S.Mc); S.Xc = upd ? Xn : (on ? DP_NEG_INF : S.Xc); S.Yc = upd ? Yn : (on ? DP_NEG_INF : S.Yc);
        } else { S.Mc = upd ? Mn : S.Mc; S.Xc = upd ? Xn : S.Xc; S.Yc = upd ? Yn : S.Yc; }
        S.Md = Mu; S.Xd = Xu; S.Yd = Yu;
    }
}

// ---- big intervals: one workgroup (DP_MW_WAVES waves) per interval ---------------------------------------------------------
// The 64-row stripes of a long profile go round-robin to the waves and run as a software pipeline: stripe s
// follows stripe s-1 three 64-step rounds behind, which is when the columns of the parked boundary row it is about to
// consume (and the chunk it prefetches) are final.  Progress is published per wave in LDS between two barriers per
// round; a wave whose dependency is not met sits the round out.  Same recurrences, same traceback bytes, same
// results as the one-wave path -- only the schedule differs.
constexpr int DP_MW_LAG = 3;
constexpr int DP_MW_WIN = 256;             // steps of traceback the walk keeps in LDS (16 KB)
constexpr int DP_MW_WAVES = 16;            // 1024 threads: four waves per SIMD of one CU

// the pipeline over the stripes of one step (all waves of the workgroup call it together)
template <bool BANDED>
__device__ __forceinline__ void dp_mw_sweep(int32_t *s_stripe, int32_t *s_round, int32_t *s_fin, int lane, int wv, int32_t m, int32_t n,
                                        int32_t nstripes, const uint32_t *Pc, const uint8_t *seq, const DpScoring &sc, int32_t krows,
                                        int32_t *rowbuf, uint8_t *tbp, int32_t T)
{
    constexpr int W = DP_MW_WAVES;
    int32_t my_s = wv, my_c = 0;
    if (lane == 0) { s_stripe[wv] = wv; s_round[wv] = 0; }
    __syncthreads();
    DpStripe S;                                       // the stripe this wave is in the middle of
    int32_t nrounds = 0;
    // every stripe runs at most (n + 64) / 64 + 1 rounds; a fully serial schedule is the upper bound
    const int64_t guard_max = (int64_t)nstripes * ((T + 64) / 64 + 2) + 16;
    for (int64_t guard = 0; guard < guard_max; guard++) {
        int32_t lo = s_stripe[0];
#pragma unroll
        for (int w = 1; w < W; w++) lo = min(lo, s_stripe[w]);
        if (lo >= nstripes) break;
        bool can = my_s < nstripes;
        if (can && my_s > 0) {
            const int pw = (my_s - 1) % W;
            const int32_t ps = s_stripe[pw], pr = s_round[pw];
            // a banded stripe starts its sweep further right than the one above: that many more of the rounds above must be done
            const int32_t dj = BANDED ? (dp_j0(my_s, m, n, true) - dp_j0(my_s - 1, m, n, true) + 63) / 64 : 0;
            can = ps > my_s - 1 || (ps == my_s - 1 && pr >= my_c + DP_MW_LAG + dj);
        }
        __syncthreads();               // every wave has read this round's state before anyone updates it
        if (can) {
            if (my_c == 0) {
                dp_stripe_begin<BANDED>(S, my_s, lane, m, n, nstripes, Pc, seq, sc, krows, rowbuf, tbp, T);
                nrounds = (S.steps + 63) / 64;
            }
            dp_stripe_round<BANDED>(S, my_c, lane);
            my_c++;
            if (my_c == nrounds) {
                if (my_s == nstripes - 1 && lane == ((m - 1) & 63)) { s_fin[0] = S.Mc; s_fin[1] = S.Xc; s_fin[2] = S.Yc; }   // the lane of row m holds (m, n)
                my_s += W; my_c = 0;
            }
            __threadfence_block();     // parked row and traceback bytes before the progress counters
            if (lane == 0) { s_stripe[wv] = my_s; s_round[wv] = my_c; }
        }
        __syncthreads();
    }
}

__device__ void dp_interval_mw(int nseq, int64_t iv, const uint8_t *__restrict__ codes, const int64_t *__restrict__ seq_off,
                               DpMeta *__restrict__ meta, uint32_t *__restrict__ cntA, uint32_t *__restrict__ maskA,
                               uint32_t *__restrict__ cntB, uint32_t *__restrict__ maskB, uint8_t *__restrict__ tb,
                               const int64_t *__restrict__ tb_off, int32_t *__restrict__ rows,
                               const int64_t *__restrict__ rows_off, uint8_t *__restrict__ ops, const DpScoring &sc, int64_t band_from)
{
    __shared__ int32_t s_stripe[DP_MW_WAVES], s_round[DP_MW_WAVES], s_fin[3];
    __shared__ __attribute__((aligned(16))) uint8_t s_win[DP_MW_WIN * 64];     // traceback window of the walk
    constexpr int W = DP_MW_WAVES;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    DpMeta mt; mt.m = 0; mt.krows = 0; mt.cur = 0; mt.pad = 0; mt.score = 0; mt.cells = 0;
    const int64_t base = seq_off[iv * nseq];
    int64_t longest = 0;
    for (int g = 0; g < nseq; g++) longest = max(longest, seq_off[iv * nseq + g + 1] - seq_off[iv * nseq + g]);
    const bool banded = longest > band_from;
    for (int g = 0; g < nseq; g++) {
        const int64_t so = seq_off[iv * nseq + g];
        const int32_t n = (int32_t)(seq_off[iv * nseq + g + 1] - so);
        if (n == 0) continue;
        const uint8_t *seq = codes + so;
        uint32_t *Pc = (mt.cur ? cntB : cntA) + base, *Pm = (mt.cur ? maskB : maskA) + base;
        uint32_t *Qc = (mt.cur ? cntA : cntB) + base, *Qm = (mt.cur ? maskA : maskB) + base;
        if (mt.krows == 0) {
            for (int32_t c = threadIdx.x; c < n; c += 64 * W) { Pc[c] = 1u << (8 * seq[c]); Pm[c] = 1u << g; }
            mt.m = n; mt.krows = 1;
            __threadfence_block();
            __syncthreads();
            continue;
        }
        const int32_t m = mt.m;
        const int32_t T = (int32_t)dp_stride(m, n, banded);
        auto j0_of = [&](int32_t st) -> int32_t { return dp_j0(st, m, n, banded); };
        uint8_t *tbp = tb + tb_off[iv];
        int32_t *rowbuf = rows + rows_off[iv];
        const int32_t nstripes = (m + 63) / 64;

        if (banded) dp_mw_sweep<true>(s_stripe, s_round, s_fin, lane, wv, m, n, nstripes, Pc, seq, sc, mt.krows, rowbuf, tbp, T);
        else dp_mw_sweep<false>(s_stripe, s_round, s_fin, lane, wv, m, n, nstripes, Pc, seq, sc, mt.krows, rowbuf, tbp, T);
        int32_t fM, fX, fY;
        fM = s_fin[0]; fX = s_fin[1]; fY = s_fin[2];
        int32_t best = fM; int state = 0;
        if (fX > best) { best = fX; state = 1; }
        if (fY > best) { best = fY; state = 2; }

        // ---- traceback: every wave walks the same path (uniform control flow), wave 0 records it ----
        uint8_t *opr = ops + base;
        // through a window in LDS that the whole workgroup refills (see the one-wave walker in dp_step)
        int32_t ti = m, tj = n, len = 0, ws = -1, wj0 = 0, wlo = 0;
        while (ti > 0 || tj > 0) {
            uint32_t op, nstate;
            if (ti == 0) { op = 2; nstate = (tj == 1) ? 0 : 2; }
            else {
                const int32_t s = (ti - 1) >> 6, l = (ti - 1) & 63;
                if (s != ws) wj0 = j0_of(s);
                const int32_t t = tj - wj0 + l;
                if (s != ws || t < wlo) {                  // the same for every wave: the barriers are uniform
                    ws = s; wlo = max(0, t - (DP_MW_WIN - 1));
                    const uint8_t *src = tbp + ((size_t)s * T + wlo) * 64;
                    const int32_t nbytes = (t - wlo + 1) * 64;
                    __syncthreads();                       // nobody still reads the old window
                    for (int32_t o = threadIdx.x * 16; o < nbytes; o += 64 * W * 16)
                        *reinterpret_cast<uint4 *>(s_win + o) = *reinterpret_cast<const uint4 *>(src + o);
                    __syncthreads();
                }
                const uint8_t bt = s_win[(size_t)(t - wlo) * 64 + l];
                if (state == 0) { op = 3; nstate = bt & 3; }
                else if (state == 1) { op = 1; nstate = (bt >> 2) & 3; }
                else { op = 2; nstate = (bt >> 4) & 3; }
            }
            if (threadIdx.x == 0) opr[len] = (uint8_t)op;
            len++;
            if (op & 1) ti--;
            if (op & 2) tj--;
            state = (int)nstate;
        }
        __threadfence_block();
        __syncthreads();
        // ---- new profile: every wave keeps the running source counts, chunk k is written by wave k mod W ----
        int32_t carry_p = 0, carry_s = 0;
        for (int32_t c0i = 0, k = 0; c0i < len; c0i += 64, k++) {
            const int32_t c = c0i + lane;
            const bool ok = c < len;
            const uint32_t op = ok ? opr[len - 1 - c] : 0u;
            const uint64_t bp = __ballot(ok && (op & 1)), bs = __ballot(ok && (op & 2));
            if (ok && (k % W) == wv) {
                const int32_t pi = carry_p + (int32_t)__popcll(bp & lt), sj = carry_s + (int32_t)__popcll(bs & lt);
                uint32_t cv = 0, mv = 0;
                if (op & 1) { cv = Pc[pi]; mv = Pm[pi]; }
                if (op & 2) { cv += 1u << (8 * seq[sj]); mv |= 1u << g; }
                Qc[c] = cv; Qm[c] = mv;
            }
            carry_p += (int32_t)__popcll(bp); carry_s += (int32_t)__popcll(bs);
        }
        mt.cells += dp_step_cells(m, n, banded, lane); mt.score += best; mt.m = len; mt.krows += 1; mt.cur ^= 1;
        __threadfence_block();
        __syncthreads();
    }
    if (threadIdx.x == 0) meta[iv] = mt;
}


// ---- small intervals: several per wave -------------------------------------------------------------------------
// Most inter-anchor intervals of a closely related genome set are a handful of bases: a profile of m <= 16 rows
// keeps 48 of a wave's 64 lanes idle in dp_step.  Here a wave is cut into 64/G groups of G lanes (G = 16 or 32) and
// every group runs its own interval through the same systolic recurrence: the DPP wave shift still moves
// (i-1, .) down the whole wave and each group's first lane overrides what it received with its own boundary row
// (always the analytic first row: one stripe); the group leader's next base comes from the group's preloaded
// chunk by ds_bpermute; traceback bytes and reversed ops live in the wave's LDS slice, cut per group; the
// traceback walks are run by the group leaders side by side.  The step loop runs to the longest group of the wave
// (the list is sorted by size, so neighbours are alike).  Same recurrences and tie rules as dp_stripe_round.
constexpr int DP_GRP_TMAX = DP_LDS_TB / 64;        // 128 systolic steps, and m + n <= 128 ops, per group

__device__ __forceinline__ int32_t lane_read(int32_t v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }

template <int G>
__device__ void dp_groups(int nseq, const int64_t *__restrict__ list, int64_t first, int64_t count, int64_t wave_index, int64_t nwaves,
                          const uint8_t *__restrict__ codes, const int64_t *__restrict__ seq_off, DpMeta *__restrict__ meta,
                          uint32_t *__restrict__ cntA, uint32_t *__restrict__ maskA, uint32_t *__restrict__ cntB,
                          uint32_t *__restrict__ maskB, uint8_t *s_tb_wave, uint8_t *s_ops_wave, const DpScoring &sc)
{
    constexpr int GROUPS = 64 / G;
    constexpr uint64_t GMASK = G == 32 ? 0xffffffffULL : 0xffffULL;
    const int lane = threadIdx.x & 63, ql = lane & (G - 1), q = lane / G, gbase = lane & ~(G - 1);
    const bool leader = ql == 0;
    const uint32_t below = (1u << ql) - 1u;                       // ql <= 31
    uint8_t *tbq = s_tb_wave + q * (DP_GRP_TMAX * G);
    uint8_t *opq = s_ops_wave + q * DP_GRP_TMAX;
    for (int64_t li0 = wave_index * GROUPS; li0 < count; li0 += nwaves * GROUPS) {
        const bool have = li0 + q < count;
        const int64_t iv = have ? list[first + li0 + q] : 0;
        DpMeta mt; mt.m = 0; mt.krows = 0; mt.cur = 0; mt.pad = 0; mt.score = 0; mt.cells = 0;
        const int64_t base = have ? seq_off[iv * nseq] : 0;
        // The class of an interval comes from an ESTIMATE of its profile lengths (dp_class_of): a group whose profile
        // outgrows its G rows, or whose step outgrows the LDS slice, gives the interval up (mt.m = -1) and the wave
        // runs it through the one-wave path afterwards (dp_step).
        bool dead = false;
        for (int g = 0; g < nseq; g++) {
            int64_t so = 0; int32_t n = 0;
            if (have && !dead) { so = seq_off[iv * nseq + g]; n = (int32_t)(seq_off[iv * nseq + g + 1] - so); }
            if (n > 0 && mt.krows > 0 && (mt.m > G || mt.m + n > DP_GRP_TMAX)) { dead = true; n = 0; }
            const uint8_t *seq = codes + so;
            uint32_t *Pc = (mt.cur ? cntB : cntA) + base, *Pm = (mt.cur ? maskB : maskA) + base;
            uint32_t *Qc = (mt.cur ? cntA : cntB) + base, *Qm = (mt.cur ? maskA : maskB) + base;
            const bool init = n > 0 && mt.krows == 0, step = n > 0 && mt.krows > 0;
            if (init) for (int32_t c = ql; c < n; c += G) { Pc[c] = 1u << (8 * seq[c]); Pm[c] = 1u << g; }
            if (__ballot(step)) {
                const int32_t m = step ? mt.m : 0, nn = step ? n : 0;
                const int32_t i = ql + 1;
                const bool active = i <= m;
                const uint32_t cn = active ? Pc[i - 1] : 0u;
                const int32_t c0 = cn & 255, c1 = (cn >> 8) & 255, c2 = (cn >> 16) & 255, c3 = cn >> 24;
                const int32_t r = c0 + c1 + c2 + c3;
                const int32_t sub0 = c0 * sc.s[0][0] + c1 * sc.s[1][0] + c2 * sc.s[2][0] + c3 * sc.s[3][0];
                const int32_t sub1 = c0 * sc.s[0][1] + c1 * sc.s[1][1] + c2 * sc.s[2][1] + c3 * sc.s[3][1];
                const int32_t sub2 = c0 * sc.s[0][2] + c1 * sc.s[1][2] + c2 * sc.s[2][2] + c3 * sc.s[3][2];
                const int32_t sub3 = c0 * sc.s[0][3] + c1 * sc.s[1][3] + c2 * sc.s[2][3] + c3 * sc.s[3][3];
                const int32_t gxo = sc.go * r, gxe = sc.ge * r, gyo = sc.go * mt.krows, gye = sc.ge * mt.krows;
                const int32_t steps = step ? nn + m : 0;                   // t = 0 .. n + m - 1
                int32_t tmax = 0;
#pragma unroll
                for (int k = 0; k < GROUPS; k++) tmax = max(tmax, __builtin_amdgcn_readlane(steps, k * G));
                int32_t Mc = DP_NEG_INF, Xc = DP_NEG_INF, Yc = DP_NEG_INF, Md = DP_NEG_INF, Xd = DP_NEG_INF, Yd = DP_NEG_INF;
                uint32_t bcur = 0;
                // chunk k of the group's sequence: lane ql holds base k*G - 1 + ql (column t = k*G + ql reads base t-1)
                auto chunk = [&](int32_t k) -> uint32_t { return nn > 0 ? (uint32_t)seq[min(max(k * G - 1 + ql, 0), nn - 1)] : 0u; };
                uint32_t sq_cur = chunk(0), sq_nxt = chunk(1);
                for (int32_t t = 0; t < tmax; t++) {
                    if (t > 0 && (t & (G - 1)) == 0) { sq_cur = sq_nxt; sq_nxt = chunk(t / G + 1); }
                    const int32_t j = t - ql;
                    int32_t Mu = wave_shr1z(Mc), Xu = wave_shr1z(Xc), Yu = wave_shr1z(Yc);      // the leaders override what they receive
                    uint32_t bnext = (uint32_t)wave_shr1z((int32_t)bcur);
                    const uint32_t b0 = (uint32_t)lane_read((int32_t)sq_cur, gbase + (t & (G - 1)));
                    const int32_t M0 = t == 0 ? 0 : DP_NEG_INF, Y0 = t == 0 ? DP_NEG_INF : gyo + (t - 1) * gye;
                    Mu = leader ? M0 : Mu; Xu = leader ? DP_NEG_INF : Xu; Yu = leader ? Y0 : Yu; bnext = leader ? b0 : bnext;
                    bcur = bnext;
                    const bool on = active && (uint32_t)j <= (uint32_t)nn, j1 = j >= 1;
                    int32_t best; uint32_t pm, px, py;
                    max3(Md, Xd, Yd, best, pm);
                    const int32_t sa = (bnext & 1) ? sub1 : sub0, sb = (bnext & 1) ? sub3 : sub2;
                    int32_t Mn = max(best + ((bnext & 2) ? sb : sa), DP_NEG_INF);
                    max3(Mu + gxo, Xu + gxe, Yu + gxo, best, px);
                    const int32_t Xn = max(best, DP_NEG_INF);
                    max3(Mc + gyo, Xc + gyo, Yc + gye, best, py);
                    int32_t Yn = max(best, DP_NEG_INF);
                    Mn = j1 ? Mn : DP_NEG_INF; Yn = j1 ? Yn : DP_NEG_INF; pm = j1 ? pm : 0u; py = j1 ? py : 0u;
                    // straight-line: the slot (t, lane) is this lane's alone and is read only for real cells; the state moves
                    // only where the lane has a cell, so the lane of row m ends up holding (m, n)
                    tbq[t * G + ql] = (uint8_t)(pm | (px << 2) | (py << 4));
                    Mc = on ? Mn : Mc; Xc = on ? Xn : Xc; Yc = on ? Yn : Yc;
                    Md = Mu; Xd = Xu; Yd = Yu;
                }
                __threadfence_block();                       // traceback bytes: written by the lanes, read by the leader
                const int owner = gbase + max(m, 1) - 1;
                const int32_t fM = lane_read(Mc, owner), fX = lane_read(Xc, owner), fY = lane_read(Yc, owner);
                int32_t best = fM; int state = 0;
                if (fX > best) { best = fX; state = 1; }
                if (fY > best) { best = fY; state = 2; }
                // ---- traceback: the group leaders walk side by side ----
                int32_t len = 0;
                if (leader && step) {
                    int32_t ti = m, tj = nn;
                    while (ti > 0 || tj > 0) {
                        uint32_t op, nstate;
                        if (ti == 0) { op = 2; nstate = (tj == 1) ? 0 : 2; }
                        else {
                            const int32_t l = ti - 1;
                            const uint8_t bt = tbq[(tj + l) * G + l];
                            if (state == 0) { op = 3; nstate = bt & 3; }
                            else if (state == 1) { op = 1; nstate = (bt >> 2) & 3; }
                            else { op = 2; nstate = (bt >> 4) & 3; }
                        }
                        opq[len] = (uint8_t)op;
                        len++;
                        if (op & 1) ti--;
                        if (op & 2) tj--;
                        state = (int)nstate;
                    }
                }
                __threadfence_block();
                len = lane_read(len, gbase);
                int32_t maxlen = 0;
#pragma unroll
                for (int k = 0; k < GROUPS; k++) maxlen = max(maxlen, __builtin_amdgcn_readlane(len, k * G));
                // ---- new profile in forward order, per group ----
                int32_t carry_p = 0, carry_s = 0;
                for (int32_t c0i = 0; c0i < maxlen; c0i += G) {
                    const int32_t c = c0i + ql;
                    const bool ok = step && c < len;
                    const uint32_t op = ok ? opq[len - 1 - c] : 0u;
                    const uint64_t bp = __ballot(ok && (op & 1)), bs = __ballot(ok && (op & 2));
                    const uint32_t gp = (uint32_t)((bp >> gbase) & GMASK), gs = (uint32_t)((bs >> gbase) & GMASK);
                    if (ok) {
                        const int32_t pi = carry_p + (int32_t)__popc(gp & below), sj = carry_s + (int32_t)__popc(gs & below);
                        uint32_t cv = 0, mv = 0;
                        if (op & 1) { cv = Pc[pi]; mv = Pm[pi]; }
                        if (op & 2) { cv += 1u << (8 * seq[sj]); mv |= 1u << g; }
                        Qc[c] = cv; Qm[c] = mv;
                    }
                    carry_p += (int32_t)__popc(gp); carry_s += (int32_t)__popc(gs);
                }
                if (step) { mt.cells += (int64_t)m * nn; mt.score += best; mt.m = len; mt.krows += 1; mt.cur ^= 1; }
            }
            if (init) { mt.m = n; mt.krows = 1; }
            __threadfence_block();       // profiles written by some lanes are read by others in the next step
        }
        if (dead) mt.m = -1;
        if (have && leader) meta[iv] = mt;
    }
}


// ================================================================================================================
// Register-blocked sweep.  The systolic kernels above give every lane ONE profile row: a step costs ~75 instructions
// whatever the lane does with it, a profile of m rows against n bases takes m + n steps, and most inter-anchor
// intervals are a handful of bases (most lanes of the wave idle in the skew).  Here a lane owns R consecutive rows: it
// takes (i0-1, j) and (i0-1, j-1) from the lane above once per step (three DPP shifts, not three per row) and runs
// down its R cells in registers -- (i-1, j) of row r+1 is what it has just computed for row r, (i-1, j-1) what row r
// held before.  A step is ~25 + 31 R instructions for R cells, the sweep takes n + ceil(m / R) steps, and a problem of
// m <= 16 rows needs 4 lanes instead of 16: sixteen intervals share a wave (G = 4), eight with m <= 32 (G = 8), four
// with m <= 64 (G = 16); anything larger gets the wave to itself in stripes of 64 R rows (dp2_interval).  The
// traceback byte of the lane's R cells of a step is one 32-bit store.  Same recurrence, same tie rules (the first of
// M, X, Y that attains the maximum), same traceback walk and profile rebuild as above: the result is bit-identical.
// ================================================================================================================
constexpr int DP2_R = 4;                         // rows per lane
constexpr int DP2_T = 64;                        // systolic steps a sub-wave group keeps in LDS: n + G <= DP2_T
constexpr int DP2_TB_DW = DP2_T * 64;            // per wave: DP2_T steps x 64 lanes x one dword (R bytes)
constexpr int DP2_REC = 16 * (16 + DP2_T);       // per wave: reversed op records of all groups (G = 4: 16 groups x (16 rows + 64))
constexpr int DP2_SEQ = 16 * (DP2_T + 4);        // per wave: the groups' sequences, each behind G bytes of padding
constexpr int DP2_WAVES = 2;                     // waves per workgroup (40 KB of LDS: four workgroups per CU)

template <int R>
struct Dp2Rows {
    int32_t M[R], X[R], Y[R];                    // column j-1 before a step, column j after it
    int32_t s0[R], s1[R], s2[R], s3[R], gxo[R], gxe[R];
    __device__ __forceinline__ void constants(const uint32_t *Pc, int32_t i0, int32_t m, const DpScoring &sc)
    {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t cn = i0 + r < m ? Pc[i0 + r] : 0u;         // rows beyond the profile: zeros (their cells feed nothing)
            const int32_t c0 = cn & 255, c1 = (cn >> 8) & 255, c2 = (cn >> 16) & 255, c3 = cn >> 24;
            const int32_t rr = c0 + c1 + c2 + c3;
            s0[r] = c0 * sc.s[0][0] + c1 * sc.s[1][0] + c2 * sc.s[2][0] + c3 * sc.s[3][0];
            s1[r] = c0 * sc.s[0][1] + c1 * sc.s[1][1] + c2 * sc.s[2][1] + c3 * sc.s[3][1];
            s2[r] = c0 * sc.s[0][2] + c1 * sc.s[1][2] + c2 * sc.s[2][2] + c3 * sc.s[3][2];
            s3[r] = c0 * sc.s[0][3] + c1 * sc.s[1][3] + c2 * sc.s[2][3] + c3 * sc.s[3][3];
            gxo[r] = sc.go * rr; gxe[r] = sc.ge * rr;
            M[r] = X[r] = Y[r] = DP_NEG_INF;
        }
    }
    // one step: the lane's R cells of column j.  (Mu, Xu, Yu) = (i0-1, j), (Md, Xd, Yd) = (i0-1, j-1), b = base j.
    // Returns the R traceback bytes (row r in byte r).
    __device__ __forceinline__ uint32_t step(int32_t Mu, int32_t Xu, int32_t Yu, int32_t Md, int32_t Xd, int32_t Yd, uint32_t b,
                                             int32_t gyo, int32_t gye, bool j1)
    {
        const bool lo = (b & 1u) != 0, hi = (b & 2u) != 0;
        uint32_t tb = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int32_t Ml = M[r], Xl = X[r], Yl = Y[r];            // (i, j-1): the diagonal input of row r + 1
            const int32_t bd = max(max(Md, Xd), Yd);
            const uint32_t pm = Md == bd ? 0u : (Xd == bd ? 1u : 2u);
            const int32_t sa = lo ? s1[r] : s0[r], sb = lo ? s3[r] : s2[r];
            int32_t Mn = max(bd + (hi ? sb : sa), DP_NEG_INF);
            const int32_t xa = Mu + gxo[r], xb = Xu + gxe[r], xc = Yu + gxo[r];
            const int32_t bx = max(max(xa, xb), xc);
            const uint32_t px = xa == bx ? 0u : (xb == bx ? 4u : 8u);
            const int32_t Xn = max(bx, DP_NEG_INF);
            const int32_t ya = Ml + gyo, yb = Xl + gyo, yc = Yl + gye;
            const int32_t by = max(max(ya, yb), yc);
            const uint32_t py = ya == by ? 0u : (yb == by ? 16u : 32u);
            int32_t Yn = max(by, DP_NEG_INF);
            Mn = j1 ? Mn : DP_NEG_INF; Yn = j1 ? Yn : DP_NEG_INF;     // column 0: only X exists
            tb |= (pm | px | py) << (8 * r);
            M[r] = Mn; X[r] = Xn; Y[r] = Yn;
            Md = Ml; Xd = Xl; Yd = Yl; Mu = Mn; Xu = Xn; Yu = Yn;
        }
        return tb;
    }
    __device__ __forceinline__ void row(int ro, int32_t &m_, int32_t &x_, int32_t &y_) const
    {
        m_ = M[0]; x_ = X[0]; y_ = Y[0];
#pragma unroll
        for (int r = 1; r < R; r++) if (ro == r) { m_ = M[r]; x_ = X[r]; y_ = Y[r]; }
    }
};

// ---- sub-wave groups: 64 / G intervals per wave, G lanes x R rows each, everything of a step in LDS ----
template <int G>
__device__ void dp2_groups(int nseq, const int64_t *__restrict__ list, int64_t first, int64_t count, int64_t wave_index, int64_t nwaves,
                           const uint8_t *__restrict__ codes, const int64_t *__restrict__ seq_off, DpMeta *__restrict__ meta,
                           uint32_t *__restrict__ cntA, uint32_t *__restrict__ maskA, uint32_t *__restrict__ cntB,
                           uint32_t *__restrict__ maskB, uint32_t *s_tb_wave, uint16_t *s_rec_wave, uint8_t *s_seq_wave, const DpScoring &sc)
{
    constexpr int R = DP2_R, GROUPS = 64 / G, ROWS = G * R;
    const int lane = threadIdx.x & 63, ql = lane & (G - 1), q = lane / G, gbase = lane & ~(G - 1);
    const bool leader = ql == 0;
    uint32_t *tbq = s_tb_wave + q * (DP2_T * G);                       // one dword per (step, lane of the group)
    const uint8_t *tbq8 = reinterpret_cast<const uint8_t *>(tbq);
    uint16_t *recq = s_rec_wave + q * (ROWS + DP2_T);
    uint8_t *seqq = s_seq_wave + q * (DP2_T + G);
    for (int64_t li0 = wave_index * GROUPS; li0 < count; li0 += nwaves * GROUPS) {
        const bool have = li0 + q < count;
        const int64_t iv = have ? list[first + li0 + q] : 0;
        DpMeta mt; mt.m = 0; mt.krows = 0; mt.cur = 0; mt.pad = 0; mt.score = 0; mt.cells = 0;
        const int64_t base = have ? seq_off[iv * nseq] : 0;
        // The class comes from an ESTIMATE of the profile lengths (DpClassEst): a group whose profile outgrows its rows, or
        // whose step outgrows the LDS slice, gives the interval up (mt.m = -1); the wave runs it alone afterwards.
        bool dead = false;
        for (int g = 0; g < nseq; g++) {
            int64_t so = 0; int32_t n = 0;
            if (have && !dead) { so = seq_off[iv * nseq + g]; n = (int32_t)(seq_off[iv * nseq + g + 1] - so); }
            if (n > 0 && mt.krows > 0 && (mt.m > ROWS || n + G > DP2_T)) { dead = true; n = 0; }
            const uint8_t *seq = codes + so;
            uint32_t *Pc = (mt.cur ? cntB : cntA) + base, *Pm = (mt.cur ? maskB : maskA) + base;
            uint32_t *Qc = (mt.cur ? cntA : cntB) + base, *Qm = (mt.cur ? maskA : maskB) + base;
            const bool init = n > 0 && mt.krows == 0, step = n > 0 && mt.krows > 0;
            if (init) for (int32_t c = ql; c < n; c += G) { Pc[c] = 1u << (8 * seq[c]); Pm[c] = 1u << g; }
            if (__ballot(step)) {
                const int32_t m = step ? mt.m : 0, nn = step ? n : 0;
                for (int32_t c = ql; c < nn; c += G) seqq[G + c] = seq[c];       // base j at seqq[G + j - 1]
                Dp2Rows<R> L;
                L.constants(Pc, ql * R, m, sc);
                const int32_t gyo = sc.go * mt.krows, gye = sc.ge * mt.krows;
                const int32_t La = (m + R - 1) / R;                               // lanes of the group that hold rows
                const bool has = ql < La;
                const int32_t steps = step ? nn + La : 0;                          // t = 0 .. n + La - 1
                int32_t tmax = 0;
#pragma unroll
                for (int k = 0; k < GROUPS; k++) tmax = max(tmax, __builtin_amdgcn_readlane(steps, k * G));
                int32_t Md = DP_NEG_INF, Xd = DP_NEG_INF, Yd = DP_NEG_INF;
                __threadfence_block();                                             // the staged sequence: written by some lanes, read by others
                const uint8_t *sp = seqq + (G - 1 - ql);                           // sp[t] = base of column j = t - ql
                uint32_t *tw = tbq + ql;
                for (int32_t t = 0; t < tmax; t++, tw += G) {
                    const int32_t j = t - ql;
                    int32_t Mu = wave_shr1z(L.M[R - 1]), Xu = wave_shr1z(L.X[R - 1]), Yu = wave_shr1z(L.Y[R - 1]);
                    // row 0 of the matrix, for the lane that holds row 1 (its j is t)
                    const int32_t M0 = t == 0 ? 0 : DP_NEG_INF, Y0 = t == 0 ? DP_NEG_INF : gyo + (t - 1) * gye;
                    Mu = leader ? M0 : Mu; Xu = leader ? DP_NEG_INF : Xu; Yu = leader ? Y0 : Yu;
                    if (has && (uint32_t)j <= (uint32_t)nn) {
                        const uint32_t b = sp[t];
                        *tw = L.step(Mu, Xu, Yu, Md, Xd, Yd, b, gyo, gye, j >= 1);
                    }
                    Md = Mu; Xd = Xu; Yd = Yu;
                }
                __threadfence_block();                       // traceback dwords: written by the lanes, read by the leader
                const int owner = gbase + (max(m, 1) - 1) / R;
                int32_t fM, fX, fY;
                L.row((max(m, 1) - 1) % R, fM, fX, fY);       // the lane of row m holds (m, n) in that row: its state stopped at column n
                fM = lane_read(fM, owner); fX = lane_read(fX, owner); fY = lane_read(fY, owner);
                int32_t best = fM; int state = 0;
                if (fX > best) { best = fX; state = 1; }
                if (fY > best) { best = fY; state = 2; }
                // ---- traceback: the group leaders walk side by side; an op is recorded with the profile column / base it consumes ----
                int32_t len = 0;
                if (leader && step) {
                    int32_t ti = m, tj = nn;
                    while (ti > 0 || tj > 0) {
                        uint32_t op, nstate;
                        if (ti == 0) { op = 2; nstate = (tj == 1) ? 0 : 2; }
                        else {
                            const int32_t l = (ti - 1) / R, r = (ti - 1) % R;
                            const uint8_t bt = tbq8[((tj + l) * G + l) * R + r];
                            if (state == 0) { op = 3; nstate = bt & 3; }
                            else if (state == 1) { op = 1; nstate = (bt >> 2) & 3; }
                            else { op = 2; nstate = (bt >> 4) & 3; }
                        }
                        recq[len] = (uint16_t)(op | ((uint32_t)(ti - 1) & 127u) << 2 | ((uint32_t)(tj - 1) & 127u) << 9);
                        len++;
                        if (op & 1) ti--;
                        if (op & 2) tj--;
                        state = (int)nstate;
                    }
                }
                __threadfence_block();
                len = lane_read(len, gbase);
                int32_t maxlen = 0;
#pragma unroll
                for (int k = 0; k < GROUPS; k++) maxlen = max(maxlen, __builtin_amdgcn_readlane(len, k * G));
                // ---- new profile in forward order, per group: column c comes from record len - 1 - c ----
                for (int32_t c = ql; c < maxlen; c += G) {
                    if (step && c < len) {
                        const uint32_t rec = recq[len - 1 - c];
                        uint32_t cv = 0, mv = 0;
                        if (rec & 1u) { const uint32_t pi = (rec >> 2) & 127u; cv = Pc[pi]; mv = Pm[pi]; }
                        if (rec & 2u) { cv += 1u << (8 * seqq[G + ((rec >> 9) & 127u)]); mv |= 1u << g; }
                        Qc[c] = cv; Qm[c] = mv;
                    }
                }
                if (step) { mt.cells += (int64_t)m * nn; mt.score += best; mt.m = len; mt.krows += 1; mt.cur ^= 1; }
            }
            if (init) { mt.m = n; mt.krows = 1; }
            __threadfence_block();       // profiles written by some lanes are read by others in the next step
        }
        if (dead) mt.m = -1;
        if (have && leader) meta[iv] = mt;
    }
}

// ---- one interval per wave: stripes of 64 R rows, traceback in HBM (walked through an LDS window) ----
// lanes a step uses: all 64 when the profile has several stripes, else the lanes that hold rows, rounded up to 4 (so that a
// step's traceback bytes are a multiple of 16); the traceback stride of a stripe is n + that many steps
__host__ __device__ __forceinline__ int64_t dp2_lanes(int64_t m) { return m > 64 * DP2_R ? 64 : (((m + DP2_R - 1) / DP2_R + 3) & ~(int64_t)3); }
__host__ __device__ __forceinline__ int64_t dp2_tb_need(int64_t m, int64_t n)
{
    const int64_t W = dp2_lanes(m), stripes = (m + 64 * DP2_R - 1) / (64 * DP2_R);
    return stripes * (n + W) * W * DP2_R;
}

template <int R>
struct Dp2Stripe {
    Dp2Rows<R> L;
    int32_t s, n, steps, gyo, gye, La;
    bool has, park;
    const uint8_t *seq; const int32_t *rin; int32_t *rout; uint8_t *tbs; int32_t rowbytes;
    int32_t Md, Xd, Yd;
    uint32_t bcur, sq_cur, sq_nxt;
    int32_t bM_cur, bX_cur, bY_cur, bM_nxt, bX_nxt, bY_nxt;
};

template <int R>
__device__ __forceinline__ void dp2_stripe_begin(Dp2Stripe<R> &S, int32_t s, int lane, int32_t m, int32_t n, int32_t nstripes, int32_t W,
                                                 const uint32_t *Pc, const uint8_t *seq, const DpScoring &sc, int32_t krows, int32_t *rowbuf,
                                                 uint8_t *tbp)
{
    S.s = s; S.n = n; S.seq = seq;
    const int32_t rows_here = min(64 * R, m - s * 64 * R);
    S.La = (rows_here + R - 1) / R;
    S.has = lane < S.La;
    S.L.constants(Pc, s * 64 * R + lane * R, m, sc);
    S.gyo = sc.go * krows; S.gye = sc.ge * krows;
    S.rin = rowbuf + (size_t)((s & 1) ^ 1) * 3 * (n + 1);      // written by stripe s-1
    S.rout = rowbuf + (size_t)(s & 1) * 3 * (n + 1);
    S.park = s + 1 < nstripes;
    S.steps = n + S.La;                                         // t = 0 .. n + La - 1
    S.rowbytes = W * R;
    S.tbs = tbp + (size_t)s * (n + W) * S.rowbytes + lane * R;
    S.Md = S.Xd = S.Yd = DP_NEG_INF;
    S.bcur = 0;
}

// chunk k of lane 0's inputs: base t-1 and the row above the stripe at column t, for t = 64k + lane
template <int R>
__device__ __forceinline__ void dp2_stripe_chunk(const Dp2Stripe<R> &S, int32_t k, int lane, uint32_t &sq, int32_t &bM, int32_t &bX, int32_t &bY)
{
    const int32_t col = 64 * k + lane;
    sq = (uint32_t)S.seq[min(max(col - 1, 0), S.n - 1)];
    if (S.s == 0) {
        bM = col == 0 ? 0 : DP_NEG_INF; bX = DP_NEG_INF;
        bY = col == 0 ? DP_NEG_INF : S.gyo + (col - 1) * S.gye;
    } else {
        const int32_t cc = min(col, S.n);
        bM = S.rin[cc]; bX = S.rin[(S.n + 1) + cc]; bY = S.rin[2 * (S.n + 1) + cc];
    }
}

template <int R>
__device__ __forceinline__ void dp2_stripe_round(Dp2Stripe<R> &S, int32_t c, int lane)
{
    if (c == 0) dp2_stripe_chunk<R>(S, 0, lane, S.sq_cur, S.bM_cur, S.bX_cur, S.bY_cur);
    else { S.sq_cur = S.sq_nxt; S.bM_cur = S.bM_nxt; S.bX_cur = S.bX_nxt; S.bY_cur = S.bY_nxt; }
    dp2_stripe_chunk<R>(S, c + 1, lane, S.sq_nxt, S.bM_nxt, S.bX_nxt, S.bY_nxt);
    const int32_t t0 = __builtin_amdgcn_readfirstlane(64 * c), t_end = __builtin_amdgcn_readfirstlane(min(64 * c + 64, S.steps));
    const bool park = __builtin_amdgcn_readfirstlane((int)S.park) != 0;
    uint8_t *tbw = S.tbs + (size_t)t0 * S.rowbytes;
    for (int32_t t = t0; t < t_end; t++, tbw += S.rowbytes) {
        const int32_t j = t - lane;
        int32_t Mu = wave_shr1z(S.L.M[R - 1]), Xu = wave_shr1z(S.L.X[R - 1]), Yu = wave_shr1z(S.L.Y[R - 1]);
        int32_t bn = wave_shr1z((int32_t)S.bcur);
        const int sel = t & 63;
        Mu = lane0_set(Mu, __builtin_amdgcn_readlane(S.bM_cur, sel));
        Xu = lane0_set(Xu, __builtin_amdgcn_readlane(S.bX_cur, sel));
        Yu = lane0_set(Yu, __builtin_amdgcn_readlane(S.bY_cur, sel));
        bn = lane0_set(bn, __builtin_amdgcn_readlane((int32_t)S.sq_cur, sel));
        S.bcur = (uint32_t)bn;
        if (S.has && (uint32_t)j <= (uint32_t)S.n) {
            const uint32_t tb = S.L.step(Mu, Xu, Yu, S.Md, S.Xd, S.Yd, (uint32_t)bn, S.gyo, S.gye, j >= 1);
            *reinterpret_cast<uint32_t *>(tbw) = tb;
            if (park && lane == 63) { S.rout[j] = S.L.M[R - 1]; S.rout[(S.n + 1) + j] = S.L.X[R - 1]; S.rout[2 * (S.n + 1) + j] = S.L.Y[R - 1]; }
        }
        S.Md = Mu; S.Xd = Xu; S.Yd = Yu;
    }
}

// all progressive steps of interval iv, by one wave; win: the wave's LDS slice (DP2_TB_DW dwords) for the traceback walk
__device__ void dp2_interval(int nseq, int64_t iv, const uint8_t *__restrict__ codes, const int64_t *__restrict__ seq_off, DpMeta *__restrict__ meta,
                             uint32_t *__restrict__ cntA, uint32_t *__restrict__ maskA, uint32_t *__restrict__ cntB, uint32_t *__restrict__ maskB,
                             uint8_t *__restrict__ tb, const int64_t *__restrict__ tb_off, int32_t *__restrict__ rows, const int64_t *__restrict__ rows_off,
                             uint8_t *__restrict__ ops, uint8_t *win, const DpScoring &sc)
{
    constexpr int R = DP2_R;
    const int lane = threadIdx.x & 63;
    const uint64_t lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    DpMeta mt; mt.m = 0; mt.krows = 0; mt.cur = 0; mt.pad = 0; mt.score = 0; mt.cells = 0;
    const int64_t base = seq_off[iv * nseq];
    for (int g = 0; g < nseq; g++) {
        const int64_t so = seq_off[iv * nseq + g];
        const int32_t n = (int32_t)(seq_off[iv * nseq + g + 1] - so);
        if (n == 0) continue;
        const uint8_t *seq = codes + so;
        uint32_t *Pc = (mt.cur ? cntB : cntA) + base, *Pm = (mt.cur ? maskB : maskA) + base;
        uint32_t *Qc = (mt.cur ? cntA : cntB) + base, *Qm = (mt.cur ? maskA : maskB) + base;
        if (mt.krows == 0) {           // first non-empty sequence becomes the profile
            for (int32_t c = lane; c < n; c += 64) { Pc[c] = 1u << (8 * seq[c]); Pm[c] = 1u << g; }
            mt.m = n; mt.krows = 1;
            __threadfence_block();      // the next step's lanes read what other lanes just wrote
            continue;
        }
        const int32_t m = mt.m;
        const int32_t W = (int32_t)dp2_lanes(m), T = n + W, rowbytes = W * R;
        uint8_t *tbp = tb + tb_off[iv];
        int32_t *rowbuf = rows + rows_off[iv];             // 2 x 3 x (n+1)
        const int32_t nstripes = (m + 64 * R - 1) / (64 * R);
        int32_t fM = DP_NEG_INF, fX = DP_NEG_INF, fY = DP_NEG_INF;   // values at (m, n)
        for (int32_t s = 0; s < nstripes; s++) {
            Dp2Stripe<R> S;
            dp2_stripe_begin<R>(S, s, lane, m, n, nstripes, W, Pc, seq, sc, mt.krows, rowbuf, tbp);
            const int32_t nrounds = (S.steps + 63) / 64;
            for (int32_t c = 0; c < nrounds; c++) dp2_stripe_round<R>(S, c, lane);
            __threadfence_block();   // the parked row / traceback bytes are read back by this wave
            if (s == nstripes - 1) S.L.row((m - 1) % R, fM, fX, fY);     // the lane of row m holds (m, n) in that row
        }
        const int owner = ((m - 1) % (64 * R)) / R;
        fM = __shfl(fM, owner); fX = __shfl(fX, owner); fY = __shfl(fY, owner);
        int32_t best = fM; int state = 0;
        if (fX > best) { best = fX; state = 1; }
        if (fY > best) { best = fY; state = 2; }
        // ---- traceback (wave-uniform walk) through a window of the last DP2_TB_DW * 4 / rowbytes steps in LDS ----
        uint8_t *opr = ops + base;                         // reversed ops, capacity m + n
        const int32_t wsteps = (DP2_TB_DW * 4) / rowbytes;
        int32_t ti = m, tj = n, len = 0, ws = -1, wlo = 0;
        while (ti > 0 || tj > 0) {
            uint32_t op, nstate;
            if (ti == 0) { op = 2; nstate = (tj == 1) ? 0 : 2; }
            else {
                const int32_t i0 = ti - 1, s = i0 / (64 * R), l = (i0 % (64 * R)) / R, r = i0 % R;
                const int32_t t = tj + l;
                if (s != ws || t < wlo) {
                    ws = s; wlo = max(0, t - (wsteps - 1));
                    const uint8_t *src = tbp + ((size_t)s * T + wlo) * rowbytes;
                    const int32_t nbytes = (t - wlo + 1) * rowbytes;
                    for (int32_t o = lane * 16; o < nbytes; o += 1024)
                        *reinterpret_cast<uint4 *>(win + o) = *reinterpret_cast<const uint4 *>(src + o);
                    __threadfence_block();             // the window is read by every lane
                }
                const uint8_t bt = win[(size_t)(t - wlo) * rowbytes + l * R + r];
                if (state == 0) { op = 3; nstate = bt & 3; }
                else if (state == 1) { op = 1; nstate = (bt >> 2) & 3; }
                else { op = 2; nstate = (bt >> 4) & 3; }
            }
            if (lane == 0) opr[len] = (uint8_t)op;
            len++;
            if (op & 1) ti--;
            if (op & 2) tj--;
            state = (int)nstate;
        }
        __threadfence_block();
        // ---- new profile in forward order: ballot prefix counts give each column its sources ----
        int32_t carry_p = 0, carry_s = 0;
        for (int32_t c0i = 0; c0i < len; c0i += 64) {
            const int32_t c = c0i + lane;
            const bool ok = c < len;
            const uint32_t op = ok ? opr[len - 1 - c] : 0u;
            const uint64_t bp = __ballot(ok && (op & 1)), bs = __ballot(ok && (op & 2));
            if (ok) {
                const int32_t pi = carry_p + (int32_t)__popcll(bp & lt), sj = carry_s + (int32_t)__popcll(bs & lt);
                uint32_t cv = 0, mv = 0;
                if (op & 1) { cv = Pc[pi]; mv = Pm[pi]; }
                if (op & 2) { cv += 1u << (8 * seq[sj]); mv |= 1u << g; }
                Qc[c] = cv; Qm[c] = mv;
            }
            carry_p += (int32_t)__popcll(bp); carry_s += (int32_t)__popcll(bs);
        }
        mt.cells += (int64_t)m * n; mt.score += best; mt.m = len; mt.krows += 1; mt.cur ^= 1;
        __threadfence_block();
    }
    if (lane == 0) meta[iv] = mt;
}

struct DpClasses { int64_t first_med, n_med, first_c, n_c, first_s32, n_s32, first_s16, n_s16; uint32_t blocks_med, blocks_c, blocks_s32; int32_t scan; };   // list = [big | one wave | G = 16 | s32 (G = 8) | s16 (G = 4)]

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) dp_step(int nseq, const int64_t *__restrict__ list, DpClasses cl, const uint8_t *__restrict__ codes,
                                               const int64_t *__restrict__ seq_off, DpMeta *__restrict__ meta,
                                               uint32_t *__restrict__ cntA, uint32_t *__restrict__ maskA,
                                               uint32_t *__restrict__ cntB, uint32_t *__restrict__ maskB,
                                               uint8_t *__restrict__ tb, const int64_t *__restrict__ tb_off,
                                               int32_t *__restrict__ rows, const int64_t *__restrict__ rows_off,
                                               uint8_t *__restrict__ ops, DpScoring sc)
{
    // Small steps (one stripe, m + n <= 128) keep their traceback bytes and reversed ops in LDS: the traceback
    // walk is a chain of dependent 1-byte loads, ~100 cycles each from LDS against >1000 from L2/HBM.
    __shared__ __attribute__((aligned(16))) uint8_t s_tb[4][DP_LDS_TB];
    __shared__ uint8_t s_ops[4][DP_LDS_OPS];
    // The list is [dp_step_big's entries | one-wave | two per wave (m <= 32) | four per wave (m <= 16)], each class
    // largest first; the block ranges follow the same order so the long ones start first.
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int per = 1; bool only_failed = false;
    int64_t pos0, pstep, pend;
    if (blockIdx.x >= cl.blocks_med) {
        const bool s32 = blockIdx.x < cl.blocks_med + cl.blocks_s32;
        const uint32_t b0 = s32 ? cl.blocks_med : cl.blocks_med + cl.blocks_s32;
        const uint32_t nb = s32 ? cl.blocks_s32 : gridDim.x - cl.blocks_med - cl.blocks_s32;
        const int64_t widx = (int64_t)(blockIdx.x - b0) * 4 + wv, nw = (int64_t)nb * 4;
        if (s32) dp_groups<32>(nseq, list, cl.first_s32, cl.n_s32, widx, nw, codes, seq_off, meta, cntA, maskA, cntB, maskB, s_tb[wv], s_ops[wv], sc);
        else dp_groups<16>(nseq, list, cl.first_s16, cl.n_s16, widx, nw, codes, seq_off, meta, cntA, maskA, cntB, maskB, s_tb[wv], s_ops[wv], sc);
        // second look at this wave's own list positions: what a group gave up is aligned below, one interval per wave
        per = s32 ? 2 : 4; only_failed = true;
        pos0 = (s32 ? cl.first_s32 : cl.first_s16) + widx * per; pstep = nw * per; pend = (s32 ? cl.first_s32 + cl.n_s32 : cl.first_s16 + cl.n_s16);
        __threadfence_block();
    } else {
        pos0 = cl.first_med + (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6); pstep = ((int64_t)cl.blocks_med * blockDim.x) >> 6;
        pend = cl.first_med + cl.n_med;
    }
    const uint64_t lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;

    for (int64_t lb = pos0; lb < pend; lb += pstep)
    for (int q = 0; q < per && lb + q < pend; q++) {
      const int64_t iv = list[lb + q];
      if (only_failed && meta[iv].m != -1) continue;
      // all progressive steps of one interval run back to back in this wave (they only depend on each other)
      DpMeta mt; mt.m = 0; mt.krows = 0; mt.cur = 0; mt.pad = 0; mt.score = 0; mt.cells = 0;
      const int64_t base = seq_off[iv * nseq];
      for (int g = 0; g < nseq; g++) {
        const int64_t so = seq_off[iv * nseq + g];
        const int32_t n = (int32_t)(seq_off[iv * nseq + g + 1] - so);
        if (n == 0) continue;
        const uint8_t *seq = codes + so;
        uint32_t *Pc = (mt.cur ? cntB : cntA) + base, *Pm = (mt.cur ? maskB : maskA) + base;
        uint32_t *Qc = (mt.cur ? cntA : cntB) + base, *Qm = (mt.cur ? maskA : maskB) + base;
        if (mt.krows == 0) {           // first non-empty sequence becomes the profile
            for (int32_t c = lane; c < n; c += 64) { Pc[c] = 1u << (8 * seq[c]); Pm[c] = 1u << g; }
            mt.m = n; mt.krows = 1;
            __threadfence_block();      // the next step's lanes read what other lanes just wrote
            continue;
        }
        const int32_t m = mt.m;
        const int32_t T = n + 64;                          // traceback stride per stripe (steps)
        const bool in_lds = m <= 64 && (size_t)(m + n) * 64 <= DP_LDS_TB && m + n <= DP_LDS_OPS;
        uint8_t *tbp = in_lds ? s_tb[wv] : tb + tb_off[iv];
        int32_t *rowbuf = rows + rows_off[iv];             // 2 x 3 x (n+1)
        const int32_t nstripes = (m + 63) / 64;
        int32_t fM = DP_NEG_INF, fX = DP_NEG_INF, fY = DP_NEG_INF;   // values at (m, n)

        for (int32_t s = 0; s < nstripes; s++) {
            DpStripe S;
            dp_stripe_begin<false>(S, s, lane, m, n, nstripes, Pc, seq, sc, mt.krows, rowbuf, tbp, T);
            const int32_t nrounds = (S.steps + 63) / 64;
            for (int32_t c = 0; c < nrounds; c++) dp_stripe_round<false>(S, c, lane);
            __threadfence_block();   // the parked row / traceback bytes are read back by this wave
            if (s == nstripes - 1) { fM = S.Mc; fX = S.Xc; fY = S.Yc; }      // the lane of row m holds (m, n)
        }
        // result lives in the lane that owns row m
        const int owner = (m - 1) & 63;
        fM = __shfl(fM, owner); fX = __shfl(fX, owner); fY = __shfl(fY, owner);
        int32_t best = fM; int state = 0;
        if (fX > best) { best = fX; state = 1; }
        if (fY > best) { best = fY; state = 2; }

        // ---- traceback (wave-uniform walk; bytes were written by this wave) ----
        uint8_t *opr = in_lds ? s_ops[wv] : ops + base;   // reversed ops, capacity m + n
        // A step whose traceback went to global memory is walked through a window in the wave's LDS slice: the walk is a
        // chain of dependent one-byte loads (m + n of them), so each would pay a full L2 round trip; inside a stripe the
        // step index t = tj + l only falls (by 1 or 2 per op), so the DP_LDS_TB / 64 steps below the current one are
        // fetched at once with 16-byte loads and serve at least half as many ops.
        int32_t ti = m, tj = n, len = 0, ws = -1, wlo = 0;
        uint8_t *win = s_tb[wv];
        while (ti > 0 || tj > 0) {
            uint32_t op, nstate;
            if (ti == 0) { op = 2; nstate = (tj == 1) ? 0 : 2; }
            else {
                const int32_t s = (ti - 1) >> 6, l = (ti - 1) & 63;
                const int32_t t = tj + l;
                uint8_t bt;
                if (in_lds) bt = win[(size_t)t * 64 + l];
                else {
                    if (s != ws || t < wlo) {
                        ws = s; wlo = max(0, t - (DP_LDS_TB / 64 - 1));
                        const uint8_t *src = tbp + ((size_t)s * T + wlo) * 64;
                        const int32_t nbytes = (t - wlo + 1) * 64;
                        for (int32_t o = lane * 16; o < nbytes; o += 1024)
                            *reinterpret_cast<uint4 *>(win + o) = *reinterpret_cast<const uint4 *>(src + o);
                        __threadfence_block();             // the window is read by every lane
                    }
                    bt = win[(size_t)(t - wlo) * 64 + l];
                }
                if (state == 0) { op = 3; nstate = bt & 3; }
                else if (state == 1) { op = 1; nstate = (bt >> 2) & 3; }
                else { op = 2; nstate = (bt >> 4) & 3; }
            }
            if (lane == 0) opr[len] = (uint8_t)op;
            len++;
            if (op & 1) ti--;
            if (op & 2) tj--;
            state = (int)nstate;
        }
        __threadfence_block();
        // ---- new profile in forward order: ballot prefix counts give each column its sources ----
        int32_t carry_p = 0, carry_s = 0;
        for (int32_t c0i = 0; c0i < len; c0i += 64) {
            const int32_t c = c0i + lane;
            const bool ok = c < len;
            const uint32_t op = ok ? opr[len - 1 - c] : 0u;
            const uint64_t bp = __ballot(ok && (op & 1)), bs = __ballot(ok && (op & 2));
            if (ok) {
                const int32_t pi = carry_p + (int32_t)__popcll(bp & lt), sj = carry_s + (int32_t)__popcll(bs & lt);
                uint32_t cv = 0, mv = 0;
                if (op & 1) { cv = Pc[pi]; mv = Pm[pi]; }
                if (op & 2) { cv += 1u << (8 * seq[sj]); mv |= 1u << g; }
                Qc[c] = cv; Qm[c] = mv;
            }
            carry_p += (int32_t)__popcll(bp); carry_s += (int32_t)__popcll(bs);
        }
        mt.cells += (int64_t)m * n; mt.score += best; mt.m = len; mt.krows += 1; mt.cur ^= 1;
        __threadfence_block();
      }
      if (lane == 0) meta[iv] = mt;
    }
}



// ================================================================================================================
// Scan-formulated sweep (one wave per interval).  An anti-diagonal schedule needs m + n dependent steps whatever the
// lanes do; the recurrence itself does not: only ONE of the two gap states depends on the cell computed just before
// it in the sweep direction.  Going column by column with the profile rows on the lanes (orientation A), M(i, j) and
// Y(i, j) read column j - 1 only -- element-wise -- and X(i, j) = max(-inf, t_i, X(i-1, j) + gxe_i) with
// t_i = max(M, Y)(i-1, j) + gxo_i is a max-plus linear recurrence along the rows: with E_i the prefix sums of gxe it is
// X(i, j) = E_i + max_{k <= i} (max(-inf, t_k) - E_k), i.e. R local steps per lane and ONE wave-wide prefix maximum (six
// DPP steps).  A step of the sweep is a whole column (64 R rows per band), the sweep takes n + 1 steps instead of
// n + 64, and a tall thin problem (a 6.7 kb insertion against 20 bases) takes 21 steps per band of 256 rows instead of
// 105 stripes of 84 anti-diagonals.  Orientation B is the mirror image for a short profile against a long sequence:
// columns on the lanes, row by row, Y the scanned state.  The orientation is chosen per progressive step by cost.
// Values, tie rules (the first of M, X, Y that attains the maximum, found by comparing the unclamped candidates) and
// the clamp at -2^29 are those of the systolic kernels: max is associative, the sums are exact in 32 bits for every
// interval the host admits here (dp3_admissible), so the traceback bytes of every reachable cell are identical.
// ================================================================================================================
template <int CTRL, int RM> __device__ __forceinline__ int32_t dpp_keep(int32_t v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, RM, 0xf, false); }
template <int CTRL, int RM> __device__ __forceinline__ int32_t dpp_zero(int32_t v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, RM, 0xf, false); }
// inclusive prefix maximum / sum over the 64 lanes: row_shr 1, 2, 4, 8 inside the rows of 16, then row_bcast:15 and :31
// (the shifted operand's "no source lane" value is the identity of max, so that the compiler folds shift and maximum into ONE v_max_i32_dpp and
// fills the hazard slot in front of it with other work; with the lane's own value there -- update_dpp(v, v, ...) -- it emitted v_mov, s_nop, v_mov_dpp,
// v_max: four issue slots per level of a chain that every line of every sweep waits for)
template <int CTRL, int RM> __device__ __forceinline__ int32_t dpp_min(int32_t v) { return __builtin_amdgcn_update_dpp(INT32_MIN, v, CTRL, RM, 0xf, false); }
__device__ __forceinline__ int32_t wave_prefix_max(int32_t v)
{
    v = max(v, dpp_min<0x111, 0xf>(v)); v = max(v, dpp_min<0x112, 0xf>(v));
    v = max(v, dpp_min<0x114, 0xf>(v)); v = max(v, dpp_min<0x118, 0xf>(v));
    v = max(v, dpp_min<0x142, 0xa>(v)); v = max(v, dpp_min<0x143, 0xc>(v));
    return v;
}
__device__ __forceinline__ int32_t wave_prefix_sum(int32_t v)
{
    v += dpp_zero<0x111, 0xf>(v); v += dpp_zero<0x112, 0xf>(v);
    v += dpp_zero<0x114, 0xf>(v); v += dpp_zero<0x118, 0xf>(v);
    v += dpp_zero<0x142, 0xa>(v); v += dpp_zero<0x143, 0xc>(v);
    return v;
}

constexpr int DP3_R = 4, DP3_BAND = 64 * DP3_R;           // rows (A) / columns (B) of one band
// rows (A) / columns (B) a lane holds: as few as cover the dimension with 64 lanes (a tiny problem keeps 4: fewer bytes)
__host__ __device__ __forceinline__ int32_t dp3_rows_per_lane(int64_t x) { return x <= 32 ? 4 : (x <= 64 ? 1 : (x <= 128 ? 2 : 4)); }
// stride of a traceback line: four bytes per lane that holds something, in 16-byte units; whole bands beyond one
__host__ __device__ __forceinline__ int64_t dp3_pad(int64_t x)
{
    if (x > DP3_BAND) return (x + DP3_BAND - 1) / DP3_BAND * DP3_BAND;
    const int64_t R = dp3_rows_per_lane(x);
    return (4 * ((x + R - 1) / R) + 15) & ~(int64_t)15;
}
// ... and its largest value over all dimensions up to x (the host sizes for a bound of the profile length)
__host__ __device__ __forceinline__ int64_t dp3_pad_bound(int64_t x) { return x <= 32 ? ((x + 15) & ~(int64_t)15) : (x <= DP3_BAND ? DP3_BAND : (x + DP3_BAND - 1) / DP3_BAND * DP3_BAND); }
__host__ __device__ __forceinline__ bool dp3_orient_b(int64_t m, int64_t n)    // cost: steps x bands
{
    return m * ((n + DP3_BAND - 1) / DP3_BAND) < (n + 1) * ((m + DP3_BAND - 1) / DP3_BAND);
}
// traceback bytes of a step whose profile has at most m rows (either orientation), and parked boundary entries
__host__ __device__ __forceinline__ int64_t dp3_tb_need(int64_t m, int64_t n)
{
    const int64_t a = (n + 1) * dp3_pad_bound(m), b = m * dp3_pad_bound(n);
    return a > b ? a : b;
}
// parked boundary entries: A keeps the last row of a band for every column, B (more than one band of columns) the last column for every row
// (the wide sweeps below choose their orientation by super-bands, so either dimension may be the parked one)
// A cluster of workgroups (below) keeps one parked line per super-band of 2048, not two by parity.
__host__ __device__ __forceinline__ int64_t dp3_rows_need(int64_t m, int64_t n)
{
    const int64_t x = m > n ? m : n, nsb = (x + 2047) / 2048;
    return 3 * (nsb > 2 ? nsb : 2) * (x + 1);
}

// column / row steps of one progressive step on ONE wave (lines x bands in the cheaper orientation): what the launch list weighs an
// interval by when it picks the ones that get a whole workgroup (the wide sweep)
__host__ __device__ __forceinline__ int64_t dp3_scan_steps(int64_t m, int64_t n)
{
    const int64_t a = (n + 1) * ((m + DP3_BAND - 1) / DP3_BAND), b = m * ((n + DP3_BAND - 1) / DP3_BAND);
    return a < b ? a : b;
}

template <int R>
struct Dp3A {
    int32_t M[R], X[R], Y[R];
    int32_t s0[R], s1[R], s2[R], s3[R], gxo[R], gxe[R], E;
    __device__ __forceinline__ void constants(const uint32_t *Pc, int32_t i0, int32_t m, const DpScoring &sc)
    {
        int32_t bsum = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t cn = i0 + r < m ? Pc[i0 + r] : 0u;
            const int32_t c0 = cn & 255, c1 = (cn >> 8) & 255, c2 = (cn >> 16) & 255, c3 = cn >> 24;
            const int32_t rr = c0 + c1 + c2 + c3;
            s0[r] = c0 * sc.s[0][0] + c1 * sc.s[1][0] + c2 * sc.s[2][0] + c3 * sc.s[3][0];
            s1[r] = c0 * sc.s[0][1] + c1 * sc.s[1][1] + c2 * sc.s[2][1] + c3 * sc.s[3][1];
            s2[r] = c0 * sc.s[0][2] + c1 * sc.s[1][2] + c2 * sc.s[2][2] + c3 * sc.s[3][2];
            s3[r] = c0 * sc.s[0][3] + c1 * sc.s[1][3] + c2 * sc.s[2][3] + c3 * sc.s[3][3];
            gxo[r] = sc.go * rr; gxe[r] = sc.ge * rr; bsum += gxe[r];
            M[r] = X[r] = Y[r] = DP_NEG_INF;
        }
        E = wave_prefix_sum(bsum);
    }
    // column j from column j - 1.  b: base of column j (wave-uniform); t?o / t?n: the row above the band at columns j - 1 / j.
    // (J1: j >= 1.  Column 0 has no M and no Y; it is peeled off the sweep's loop so that the other columns do not pay eight selects for it.)
    template <bool J1>
    __device__ __forceinline__ uint32_t step(uint32_t b, int32_t gyo, int32_t gye, int32_t tMo, int32_t tXo, int32_t tYo,
                                             int32_t tMn, int32_t tXn, int32_t tYn)
    {
        constexpr bool j1 = J1;
        int32_t Md = lane0_set(wave_shr1z(M[R - 1]), tMo), Xd = lane0_set(wave_shr1z(X[R - 1]), tXo), Yd = lane0_set(wave_shr1z(Y[R - 1]), tYo);
        const bool lo = (b & 1u) != 0, hi = (b & 2u) != 0;
        int32_t Mn[R], Yn[R];
        uint32_t tb = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int32_t bd = max(max(Md, Xd), Yd);
            const uint32_t pm = Md == bd ? 0u : (Xd == bd ? 1u : 2u);
            const int32_t sa = lo ? s1[r] : s0[r], sb = lo ? s3[r] : s2[r];
            const int32_t mv = max(bd + (hi ? sb : sa), DP_NEG_INF);
            const int32_t ya = M[r] + gyo, yb = X[r] + gyo, yc = Y[r] + gye;
            const int32_t by = max(max(ya, yb), yc);
            const uint32_t py = ya == by ? 0u : (yb == by ? 16u : 32u);
            Mn[r] = j1 ? mv : DP_NEG_INF; Yn[r] = j1 ? max(by, DP_NEG_INF) : DP_NEG_INF;
            tb |= (pm | py) << (8 * r);
            Md = M[r]; Xd = X[r]; Yd = Y[r];
        }
        // the scanned state: local chains, one prefix maximum over the lanes, then the cells with their real carry-in
        const int32_t Mu0 = lane0_set(wave_shr1z(Mn[R - 1]), tMn), Yu0 = lane0_set(wave_shr1z(Yn[R - 1]), tYn);
        int32_t a = DP_NEG_INF, Mu = Mu0, Yu = Yu0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int32_t t = max(Mu, Yu) + gxo[r];
            a = r == 0 ? max(t, DP_NEG_INF) : max(max(t, a + gxe[r]), DP_NEG_INF);
            Mu = Mn[r]; Yu = Yn[r];
        }
        const int32_t pv = wave_prefix_max(a - E);
        const int32_t xout = max(E + pv, tXn + E);
        int32_t Xu = lane0_set(wave_shr1z(xout), tXn);
        Mu = Mu0; Yu = Yu0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int32_t xa = Mu + gxo[r], xb = Xu + gxe[r], xc = Yu + gxo[r];
            const int32_t bx = max(max(xa, xb), xc);
            const uint32_t px = xa == bx ? 0u : (xb == bx ? 4u : 8u);
            tb |= px << (8 * r);
            Xu = max(bx, DP_NEG_INF);
            X[r] = Xu; Mu = Mn[r]; Yu = Yn[r]; M[r] = Mn[r]; Y[r] = Yn[r];
        }
        return tb;
    }
};

template <int R>
struct Dp3B {
    int32_t M[R], X[R], Y[R];
    uint32_t bases;                                        // 2 bits per column of the lane
    int32_t E;
    // row i from row i - 1.  s0..s3, gxo, gxe: the profile row (wave-uniform); l?o / l?n: the column left of the band at rows i - 1 / i.
    __device__ __forceinline__ uint32_t step(int32_t s0, int32_t s1, int32_t s2, int32_t s3, int32_t gxo, int32_t gxe, int32_t gyo, int32_t gye,
                                             int32_t lMo, int32_t lXo, int32_t lYo, int32_t lMn, int32_t lXn, int32_t lYn)
    {
        int32_t Md = lane0_set(wave_shr1z(M[R - 1]), lMo), Xd = lane0_set(wave_shr1z(X[R - 1]), lXo), Yd = lane0_set(wave_shr1z(Y[R - 1]), lYo);
        int32_t Mn[R], Xn[R];
        uint32_t tb = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int32_t bd = max(max(Md, Xd), Yd);
            const uint32_t pm = Md == bd ? 0u : (Xd == bd ? 1u : 2u);
            const uint32_t b = (bases >> (2 * r)) & 3u;
            const int32_t sa = (b & 1u) ? s1 : s0, sb = (b & 1u) ? s3 : s2;
            Mn[r] = max(bd + ((b & 2u) ? sb : sa), DP_NEG_INF);
            const int32_t xa = M[r] + gxo, xb = X[r] + gxe, xc = Y[r] + gxo;
            const int32_t bx = max(max(xa, xb), xc);
            const uint32_t px = xa == bx ? 0u : (xb == bx ? 4u : 8u);
            Xn[r] = max(bx, DP_NEG_INF);
            tb |= (pm | px) << (8 * r);
            Md = M[r]; Xd = X[r]; Yd = Y[r];
        }
        const int32_t Ml0 = lane0_set(wave_shr1z(Mn[R - 1]), lMn), Xl0 = lane0_set(wave_shr1z(Xn[R - 1]), lXn);
        int32_t a = DP_NEG_INF, Ml = Ml0, Xl = Xl0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int32_t t = max(Ml, Xl) + gyo;
            a = r == 0 ? max(t, DP_NEG_INF) : max(max(t, a + gye), DP_NEG_INF);
            Ml = Mn[r]; Xl = Xn[r];
        }
        const int32_t pv = wave_prefix_max(a - E);
        const int32_t yout = max(E + pv, lYn + E);
        int32_t Yl = lane0_set(wave_shr1z(yout), lYn);
        Ml = Ml0; Xl = Xl0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int32_t ya = Ml + gyo, yb = Xl + gyo, yc = Yl + gye;
            const int32_t by = max(max(ya, yb), yc);
            const uint32_t py = ya == by ? 0u : (yb == by ? 16u : 32u);
            tb |= py << (8 * r);
            Yl = max(by, DP_NEG_INF);
            Y[r] = Yl; Ml = Mn[r]; Xl = Xn[r]; M[r] = Mn[r]; X[r] = Xn[r];
        }
        return tb;
    }
};

// Traceback of one step, all lanes together: from (ti, tj) in `state` the wave looks 64 cells ahead along the state's own
// direction (M: the diagonal, X: up, Y: left) -- lane l reads the byte of cell l -- and a ballot tells how long the run of
// "the predecessor is the same state again" is: a gap of 6.7 kb is a hundred ballots, not 6.7 k dependent loads.  The bytes
// come from a rectangle of the matrix kept in LDS (the whole step when it fits).  ops: reversed op bytes (global).
// Layout of a step's traceback: the dimension on the lanes (A: rows, B: columns) is stored lane by lane, four bytes per lane
// of which the first R are used: index x (0-based) sits at byte P(x) = (x / R) * 4 + x % R of its line; a line per step of the
// sweep (A: per column j, B: per row i).
struct Dp3Walk {
    const uint8_t *tb; int64_t stride; bool orient_b; int32_t R;
    uint8_t *win; int32_t cap;                                   // LDS window
    int32_t p_lo, p_hi, s_lo, s_hi, PW;                           // lane-dimension bytes [p_lo, p_hi] x lines [s_lo, s_hi] are in the window (p_hi < p_lo: empty)
    __device__ __forceinline__ int32_t P(int32_t x) const { return R == 4 ? x : (R == 2 ? ((x >> 1) << 2) + (x & 1) : (R == 1 ? x << 2 : (x / 3) * 4 + x % 3)); }
    // rectangle with the cell (lane-dimension index x, line t) at its lower right corner
    __device__ __forceinline__ void load(int32_t x, int32_t t, int32_t tmin, int lane)
    {
        const int32_t pend = (P(x) + 16) & ~15;                  // exclusive end, multiple of 16
        int32_t pw = min(pend, 256);
        int32_t nl = min(t - tmin + 1, cap / pw);
        if (nl < 64 && t - tmin + 1 > nl) { pw = min(pend, 128); nl = min(t - tmin + 1, cap / pw); }     // square-ish: look further back along the lines
        PW = pw; p_lo = pend - pw; p_hi = pend - 1; s_hi = t; s_lo = t - nl + 1;
        const int32_t per = pw / 16, total = per * nl;
        for (int32_t u0 = lane; u0 < total; u0 += 256) {         // four 16-byte loads in flight per lane
            uint4 v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int32_t u = min(u0 + 64 * q, total - 1), cl = u / per, seg = u % per;
                v[q] = *reinterpret_cast<const uint4 *>(tb + (size_t)(s_lo + cl) * stride + p_lo + seg * 16);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int32_t u = u0 + 64 * q, cl = u / per, seg = u % per;
                if (u < total) *reinterpret_cast<uint4 *>(win + (size_t)cl * pw + seg * 16) = v[q];
            }
        }
        __threadfence_block();
    }
    __device__ __forceinline__ bool inside(int32_t x, int32_t t) const { const int32_t p = P(x); return p >= p_lo && p <= p_hi && t >= s_lo && t <= s_hi; }
    __device__ __forceinline__ uint8_t at(int32_t x, int32_t t) const { return win[(size_t)(t - s_lo) * PW + (P(x) - p_lo)]; }
};

// walks from (m, n) in `state` to (0, 0); returns the number of ops written (reversed) to opr
__device__ __forceinline__ int32_t dp3_walk(Dp3Walk &W, int32_t m, int32_t n, int state, uint8_t *opr, int lane)
{
    int32_t ti = m, tj = n, len = 0;
    W.p_lo = 1; W.p_hi = 0; W.s_lo = 1; W.s_hi = 0;
    const bool ob = W.orient_b;
    while (ti > 0 || tj > 0) {
        if (ti == 0) {                                   // row 0: only Y exists
            for (int32_t o = lane; o < tj; o += 64) opr[len + o] = 2;
            len += tj; tj = 0;
            break;
        }
        if (tj == 0) {                                   // column 0: only X exists (the predecessor is X again, or M at row 1: the op is the same)
            for (int32_t o = lane; o < ti; o += 64) opr[len + o] = 1;
            len += ti; ti = 0;
            break;
        }
        // A: cell (i, j) = (lane-dimension index i - 1, line j); B: (j - 1, line i - 1)
        if (!(ob ? W.inside(tj - 1, ti - 1) : W.inside(ti - 1, tj))) { if (ob) W.load(tj - 1, ti - 1, 0, lane); else W.load(ti - 1, tj, 0, lane); }
        const int32_t di = state == 2 ? 0 : 1, dj = state == 1 ? 0 : 1;
        const int32_t ci = ti - lane * di, cj = tj - lane * dj;
        const bool valid = ci >= 1 && cj >= 1 && (ob ? W.inside(cj - 1, ci - 1) : W.inside(ci - 1, cj));    // (row 0 / column 0 end the look-ahead: handled on arrival)
        uint32_t ns = 3;
        if (valid) { const uint32_t bt = ob ? W.at(cj - 1, ci - 1) : W.at(ci - 1, cj); ns = (bt >> (2 * state)) & 3u; }
        const uint64_t cont = __ballot(valid && ns == (uint32_t)state);
        const int32_t k = cont == ~0ULL ? 64 : (int32_t)__builtin_ctzll(~cont);     // lanes 0 .. k-1 continue in the same state
        // lane k (if it looked at a real cell) is the cell where the state changes: its op still belongs to the run
        const uint64_t vmask = __ballot(valid);
        const bool turn = k < 64 && ((vmask >> k) & 1ULL);
        const int32_t cnt = turn ? k + 1 : k;            // >= 1: lane 0 is always valid here
        const uint8_t op = state == 0 ? 3 : (state == 1 ? 1 : 2);
        if (lane < cnt) opr[len + lane] = op;
        len += cnt;
        if (turn) state = (int)__builtin_amdgcn_readlane((int32_t)ns, k);
        ti -= cnt * di; tj -= cnt * dj;
    }
    return len;
}

// 32-bit exactness of the scan: the prefix sums of the gap-extension terms (|E| <= len * rows * |extend|) must stay far from
// the clamp; otherwise the interval keeps the anti-diagonal kernel
__host__ __device__ __forceinline__ bool dp3_admissible(int64_t total_len, int64_t krows_max, int64_t ge, int64_t go)
{
    const int64_t a = ge < 0 ? -ge : ge, b = go < 0 ? -go : go;
    return total_len * krows_max * (a > b ? a : b) < (1LL << 28) && ge <= 0 && go <= 0;
}

// orientation A, all bands: rows on the lanes (R per lane, bands of 64 R), column by column.  Leaves (m, n) in fM / fX / fY.
template <int R>
__device__ __forceinline__ void dp3_sweep_a(int lane, int32_t m, int32_t n, const uint32_t *Pc, const uint8_t *seq, const DpScoring &sc, int32_t krows,
                                            int32_t *rowbuf, uint8_t *tbp, int32_t mpad, int32_t &fM, int32_t &fX, int32_t &fY)
{
    constexpr int BAND = 64 * R;
    const int32_t nbands = (m + BAND - 1) / BAND;
    const int32_t gyo = sc.go * krows, gye = sc.ge * krows;
    for (int32_t s = 0; s < nbands; s++) {
        Dp3A<R> L;
        const int32_t i0 = s * BAND + lane * R;                      // 0-based first row of the lane
        L.constants(Pc, i0, m, sc);
        const int32_t *rin = rowbuf + (size_t)((s & 1) ^ 1) * 3 * (n + 1);
        int32_t *rout = rowbuf + (size_t)(s & 1) * 3 * (n + 1);
        const bool park = s + 1 < nbands, writes = (s * 64 + lane) * 4 < mpad;
        // What a column needs from outside the band -- its base, and (bands below the first) the parked row of the band above --
        // is fetched by the whole wave 64 columns at a time, one chunk ahead, and handed out with v_readlane: a load per
        // column would put a memory round trip into every step.  Chunk k: lane l holds column 64 k + l.
        auto chunk = [&](int32_t k, uint32_t &sq, int32_t &cM, int32_t &cX, int32_t &cY) {
            const int32_t col = 64 * k + lane;
            sq = (uint32_t)seq[min(max(col - 1, 0), n - 1)];
            if (s == 0) { cM = col == 0 ? 0 : DP_NEG_INF; cX = DP_NEG_INF; cY = col == 0 ? DP_NEG_INF : gyo + (col - 1) * gye; }
            else { const int32_t cc = min(col, n); cM = rin[cc]; cX = rin[(n + 1) + cc]; cY = rin[2 * (n + 1) + cc]; }
        };
        uint32_t sq_cur, sq_nxt; int32_t cM_cur, cX_cur, cY_cur, cM_nxt, cX_nxt, cY_nxt;
        chunk(0, sq_nxt, cM_nxt, cX_nxt, cY_nxt);
        int32_t tMo = DP_NEG_INF, tXo = DP_NEG_INF, tYo = DP_NEG_INF;
        uint8_t *tw = tbp + (size_t)(s * 64 + lane) * 4;
        for (int32_t k = 0; 64 * k <= n; k++) {
            sq_cur = sq_nxt; cM_cur = cM_nxt; cX_cur = cX_nxt; cY_cur = cY_nxt;
            chunk(k + 1, sq_nxt, cM_nxt, cX_nxt, cY_nxt);
            const int32_t jend = min(64 * k + 63, n);
            auto line = [&](int32_t j, auto first) {
                const int sel = j & 63;
                const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int32_t)sq_cur, sel);
                const int32_t tMn = __builtin_amdgcn_readlane(cM_cur, sel), tXn = __builtin_amdgcn_readlane(cX_cur, sel),
                              tYn = __builtin_amdgcn_readlane(cY_cur, sel);
                const uint32_t tbw = L.template step<!decltype(first)::value>(b, gyo, gye, tMo, tXo, tYo, tMn, tXn, tYn);
                if (writes) *reinterpret_cast<uint32_t *>(tw) = tbw;
                if (park && lane == 63) { rout[j] = L.M[R - 1]; rout[(n + 1) + j] = L.X[R - 1]; rout[2 * (n + 1) + j] = L.Y[R - 1]; }
                tMo = tMn; tXo = tXn; tYo = tYn;
            };
            int32_t j = 64 * k;
            if (k == 0) { line(0, std::true_type()); j = 1; tw += mpad; }           // column 0, peeled
            for (; j <= jend; j++, tw += mpad) line(j, std::false_type());
        }
        __threadfence_block();
        if (s == nbands - 1) {
            int32_t a = L.M[0], b2 = L.X[0], c2 = L.Y[0];
#pragma unroll
            for (int r = 1; r < R; r++) if (((m - 1) % R) == r) { a = L.M[r]; b2 = L.X[r]; c2 = L.Y[r]; }
            const int owner = ((m - 1) % BAND) / R;
            fM = __shfl(a, owner); fX = __shfl(b2, owner); fY = __shfl(c2, owner);
        }
    }
}

// orientation B, all bands: columns on the lanes, row by row
template <int R>
__device__ __forceinline__ void dp3_sweep_b(int lane, int32_t m, int32_t n, const uint32_t *Pc, const uint8_t *seq, const DpScoring &sc, int32_t krows,
                                            int32_t *rowbuf, uint8_t *tbp, int32_t npad, int32_t &fM, int32_t &fX, int32_t &fY)
{
    constexpr int BAND = 64 * R;
    const int32_t nbands = (n + BAND - 1) / BAND;
    const int32_t gyo = sc.go * krows, gye = sc.ge * krows;
    for (int32_t s = 0; s < nbands; s++) {
        Dp3B<R> L;
        const int32_t j0 = s * BAND + lane * R;                      // 0-based first column index (j - 1) of the lane
        L.bases = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            L.bases |= (uint32_t)(j0 + r < n ? seq[j0 + r] : 0) << (2 * r);
            // row 0: only Y exists (analytic, not clamped: the systolic kernels' boundary row)
            L.M[r] = DP_NEG_INF; L.X[r] = DP_NEG_INF; L.Y[r] = gyo + (j0 + r) * gye;
        }
        L.E = (lane + 1) * R * gye;
        const int32_t *cin = rowbuf + (size_t)((s & 1) ^ 1) * 3 * (m + 1);
        int32_t *cout = rowbuf + (size_t)(s & 1) * 3 * (m + 1);
        const bool park = s + 1 < nbands, writes = (s * 64 + lane) * 4 < npad;
        // per row from outside the band: the profile column's counts and (bands right of the first) the parked column of the
        // band to the left -- fetched 64 rows at a time, one chunk ahead (see orientation A).  Chunk k: lane l holds row 64 k + l + 1.
        auto chunk = [&](int32_t k, uint32_t &pc, int32_t &cM, int32_t &cX, int32_t &cY) {
            const int32_t row = min(64 * k + lane + 1, m);
            pc = Pc[row - 1];
            if (s == 0) { cM = cX = cY = DP_NEG_INF; }
            else { cM = cin[row]; cX = cin[(m + 1) + row]; cY = cin[2 * (m + 1) + row]; }
        };
        // column 0 (band 0): M and Y do not exist there, X is the chain down from (0, 0)
        int32_t lMo, lXo, lYo;
        if (s == 0) { lMo = 0; lXo = DP_NEG_INF; lYo = DP_NEG_INF; }
        else { lMo = __builtin_amdgcn_readfirstlane(cin[0]); lXo = __builtin_amdgcn_readfirstlane(cin[(m + 1)]); lYo = __builtin_amdgcn_readfirstlane(cin[2 * (m + 1)]); }
        if (park && lane == 63) { cout[0] = L.M[R - 1]; cout[(m + 1)] = L.X[R - 1]; cout[2 * (m + 1)] = L.Y[R - 1]; }    // row 0 of the band's last column
        uint32_t pc_cur, pc_nxt; int32_t cM_cur, cX_cur, cY_cur, cM_nxt, cX_nxt, cY_nxt;
        chunk(0, pc_nxt, cM_nxt, cX_nxt, cY_nxt);
        uint8_t *tw = tbp + (size_t)(s * 64 + lane) * 4;
        for (int32_t k = 0; 64 * k < m; k++) {
            pc_cur = pc_nxt; cM_cur = cM_nxt; cX_cur = cX_nxt; cY_cur = cY_nxt;
            chunk(k + 1, pc_nxt, cM_nxt, cX_nxt, cY_nxt);
            const int32_t iend = min(64 * k + 64, m);
            for (int32_t i = 64 * k + 1; i <= iend; i++, tw += npad) {
                const int sel = (i - 1) & 63;
                const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int32_t)pc_cur, sel);
                const int32_t c0 = c & 255, c1 = (c >> 8) & 255, c2 = (c >> 16) & 255, c3 = c >> 24, rr = c0 + c1 + c2 + c3;
                const int32_t s0 = c0 * sc.s[0][0] + c1 * sc.s[1][0] + c2 * sc.s[2][0] + c3 * sc.s[3][0];
                const int32_t s1 = c0 * sc.s[0][1] + c1 * sc.s[1][1] + c2 * sc.s[2][1] + c3 * sc.s[3][1];
                const int32_t s2 = c0 * sc.s[0][2] + c1 * sc.s[1][2] + c2 * sc.s[2][2] + c3 * sc.s[3][2];
                const int32_t s3 = c0 * sc.s[0][3] + c1 * sc.s[1][3] + c2 * sc.s[2][3] + c3 * sc.s[3][3];
                const int32_t gxo = sc.go * rr, gxe = sc.ge * rr;
                int32_t lMn, lXn, lYn;
                if (s == 0) {
                    const int32_t xa = lMo + gxo, xb = lXo + gxe, xc = lYo + gxo, bx = max(max(xa, xb), xc);
                    lMn = DP_NEG_INF; lYn = DP_NEG_INF; lXn = max(bx, DP_NEG_INF);
                } else { lMn = __builtin_amdgcn_readlane(cM_cur, sel); lXn = __builtin_amdgcn_readlane(cX_cur, sel); lYn = __builtin_amdgcn_readlane(cY_cur, sel); }
                const uint32_t tbw = L.step(s0, s1, s2, s3, gxo, gxe, gyo, gye, lMo, lXo, lYo, lMn, lXn, lYn);
                if (writes) *reinterpret_cast<uint32_t *>(tw) = tbw;
                if (park && lane == 63) { cout[i] = L.M[R - 1]; cout[(m + 1) + i] = L.X[R - 1]; cout[2 * (m + 1) + i] = L.Y[R - 1]; }
                lMo = lMn; lXo = lXn; lYo = lYn;
            }
        }
        __threadfence_block();
        if (s == nbands - 1) {
            int32_t a = L.M[0], b2 = L.X[0], c2 = L.Y[0];
#pragma unroll
            for (int r = 1; r < R; r++) if (((n - 1) % R) == r) { a = L.M[r]; b2 = L.X[r]; c2 = L.Y[r]; }
            const int owner = ((n - 1) % BAND) / R;
            fM = __shfl(a, owner); fX = __shfl(b2, owner); fY = __shfl(c2, owner);
        }
    }
}

__device__ void dp3_interval(int nseq, int64_t iv, const uint8_t *__restrict__ codes, const int64_t *__restrict__ seq_off, DpMeta *__restrict__ meta,
                             uint32_t *__restrict__ cntA, uint32_t *__restrict__ maskA, uint32_t *__restrict__ cntB, uint32_t *__restrict__ maskB,
                             uint8_t *__restrict__ tb, const int64_t *__restrict__ tb_off, int32_t *__restrict__ rows, const int64_t *__restrict__ rows_off,
                             uint8_t *__restrict__ ops, uint8_t *win, const DpScoring &sc)
{
    const int lane = threadIdx.x & 63;
    const uint64_t lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    DpMeta mt; mt.m = 0; mt.krows = 0; mt.cur = 0; mt.pad = 0; mt.score = 0; mt.cells = 0;
    const int64_t base = seq_off[iv * nseq];
    for (int g = 0; g < nseq; g++) {
        const int64_t so = seq_off[iv * nseq + g];
        const int32_t n = (int32_t)(seq_off[iv * nseq + g + 1] - so);
        if (n == 0) continue;
        const uint8_t *seq = codes + so;
        uint32_t *Pc = (mt.cur ? cntB : cntA) + base, *Pm = (mt.cur ? maskB : maskA) + base;
        uint32_t *Qc = (mt.cur ? cntA : cntB) + base, *Qm = (mt.cur ? maskA : maskB) + base;
        if (mt.krows == 0) {           // first non-empty sequence becomes the profile
            for (int32_t c = lane; c < n; c += 64) { Pc[c] = 1u << (8 * seq[c]); Pm[c] = 1u << g; }
            mt.m = n; mt.krows = 1;
            __threadfence_block();
            continue;
        }
        const int32_t m = mt.m;
        uint8_t *tbp = tb + tb_off[iv];
        int32_t *rowbuf = rows + rows_off[iv];
        const bool ob = dp3_orient_b(m, n);
        int32_t fM = DP_NEG_INF, fX = DP_NEG_INF, fY = DP_NEG_INF;   // values at (m, n)
        // rows (A) / columns (B) per lane: as few as cover the dimension with the 64 lanes, at most 4 (then bands)
        const int32_t ldim = ob ? n : m, R = dp3_rows_per_lane(ldim), pad = (int32_t)dp3_pad(ldim);
        Dp3Walk W; W.win = win; W.cap = DP2_TB_DW * 4; W.orient_b = ob; W.R = R; W.stride = pad;
        W.tb = ob ? tbp : tbp;                             // A: line j at tbp + j * pad; B: line i - 1 at tbp + (i - 1) * pad
        if (!ob) {
            if (R == 1) dp3_sweep_a<1>(lane, m, n, Pc, seq, sc, mt.krows, rowbuf, tbp, pad, fM, fX, fY);
            else if (R == 2) dp3_sweep_a<2>(lane, m, n, Pc, seq, sc, mt.krows, rowbuf, tbp, pad, fM, fX, fY);
            else dp3_sweep_a<4>(lane, m, n, Pc, seq, sc, mt.krows, rowbuf, tbp, pad, fM, fX, fY);
        } else {
            if (R == 1) dp3_sweep_b<1>(lane, m, n, Pc, seq, sc, mt.krows, rowbuf, tbp, pad, fM, fX, fY);
            else if (R == 2) dp3_sweep_b<2>(lane, m, n, Pc, seq, sc, mt.krows, rowbuf, tbp, pad, fM, fX, fY);
            else dp3_sweep_b<4>(lane, m, n, Pc, seq, sc, mt.krows, rowbuf, tbp, pad, fM, fX, fY);
        }
        int32_t best = fM; int state = 0;
        if (fX > best) { best = fX; state = 1; }
        if (fY > best) { best = fY; state = 2; }
        uint8_t *opr = ops + base;                         // reversed ops, capacity m + n
        const int32_t len = dp3_walk(W, m, n, state, opr, lane);
        __threadfence_block();
        // ---- new profile in forward order: ballot prefix counts give each column its sources ----
        int32_t carry_p = 0, carry_s = 0;
        for (int32_t c0i = 0; c0i < len; c0i += 64) {
            const int32_t c = c0i + lane;
            const bool ok = c < len;
            const uint32_t op = ok ? opr[len - 1 - c] : 0u;
            const uint64_t bp = __ballot(ok && (op & 1)), bs = __ballot(ok && (op & 2));
            if (ok) {
                const int32_t pi = carry_p + (int32_t)__popcll(bp & lt), sj = carry_s + (int32_t)__popcll(bs & lt);
                uint32_t cv = 0, mv = 0;
                if (op & 1) { cv = Pc[pi]; mv = Pm[pi]; }
                if (op & 2) { cv += 1u << (8 * seq[sj]); mv |= 1u << g; }
                Qc[c] = cv; Qm[c] = mv;
            }
            carry_p += (int32_t)__popcll(bp); carry_s += (int32_t)__popcll(bs);
        }
        mt.cells += (int64_t)m * n; mt.score += best; mt.m = len; mt.krows += 1; mt.cur ^= 1;
        __threadfence_block();
    }
    if (lane == 0) meta[iv] = mt;
}


// ================================================================================================================
// Wide sweep: ONE interval over the 16 waves of a workgroup, all of them on the same column (A) / row (B) at the same time.
// A single wave takes bands(m) x (n + 1) dependent column steps for a step of the progressive alignment (dp3_sweep_a: the bands
// one after the other); the largest interval of a launch -- 1 600 x 1 600 at C5, a 6.7 kb insertion in five of eight genomes at
// C4 -- then IS the launch (C5: 4.2 of 5.8 ms for one interval of 302 000).  The scan formulation has no such chain across the
// bands: M and the element-wise gap state of column j read column j - 1 only, and the scanned state is a max-plus prefix over
// ALL rows, whichever wave holds them.  So a super-band of 16 x 256 rows advances one column per step: every wave computes its
// element-wise cells and publishes its last row (barrier), every wave scans its rows with E the prefix sums over the whole
// super-band and publishes its aggregate max(a - E) (barrier), every wave folds the aggregates of the waves above into its
// carry -- which also is, plus a constant, the scanned value of the row above its first one, so nothing else has to cross --
// and finishes its cells.  Two barriers per column, no dependent chain across waves.  The arithmetic is dp3's (same clamp, same
// tie rules; E stays far from the clamp for every interval dp3_admissible lets in), the traceback layout is dp3's with R = 4
// (wave w of super-band q is band 16 q + w), so dp3_walk reads it; steps small in both dimensions run the one-wave sweep.
// ================================================================================================================
constexpr int DPW_MAXW = 16;                      // waves of a workgroup (two shapes are built: 8 waves x 4 rows per lane, 16 x 2; both super-bands hold 2048 rows)
constexpr int DPW_SB = 2048;
struct DpwShared {
    int32_t botM[DPW_MAXW], botE[DPW_MAXW];      // last cell of every wave in the current line: M and the element-wise gap state (A: Y, B: X)
    int32_t agg[DPW_MAXW];                        // max over the wave's cells of (local scan value - E)
    int32_t esum[DPW_MAXW];                       // A: sum of the gap-extension terms of the wave's rows
    int32_t fin[3];
    int32_t len;
};
__host__ __device__ __forceinline__ bool dpw_orient_b(int64_t m, int64_t n)      // cost: lines x super-bands; ties: the longer dimension on the lanes
{
    const int64_t a = (n + 1) * ((m + DPW_SB - 1) / DPW_SB), b = m * ((n + DPW_SB - 1) / DPW_SB);
    return b < a || (b == a && n > m);
}
// maximum over the waves above `wv` of their aggregates (wave-uniform result); DP_NEG_INF when there is none
__device__ __forceinline__ int32_t dpw_carry(const int32_t *agg, int wv, int lane)
{
    int32_t v = (lane & 15) < wv ? agg[lane & 15] : DP_NEG_INF;
    v = max(v, dpp_keep<0x111, 0xf>(v)); v = max(v, dpp_keep<0x112, 0xf>(v));
    v = max(v, dpp_keep<0x114, 0xf>(v)); v = max(v, dpp_keep<0x118, 0xf>(v));      // lane 15: maximum of lanes 0 .. 15
    return __builtin_amdgcn_readlane(v, 15);
}

// ---- several workgroups on ONE interval (a cluster): the super-bands of a step go round-robin to the K workgroups and run as a pipeline ----
// One CU does about 2 GCUPS of this recurrence whatever the schedule (the sweep is bound by vector issue), so an interval of several super-bands
// -- C5's largest: 560 x 4 573, three super-bands of columns, 1.8 ms of a 1.9 ms stage -- can only get faster on several CUs.  Super-band
// q + 1 needs from super-band q what a single workgroup parks between them anyway: its last row / column, line by line.  So workgroup q mod K takes
// super-band q, parks its boundary line as before, and PUBLISHES how far it is every 64 lines (all its stores drained, an agent-scope release,
// then a relaxed agent-scope store of a monotonic token: step, super-band, lines); the workgroup of q + 1 polls the token before it fetches the
// next 64 boundary values (relaxed agent-scope loads, then an agent-scope acquire: MI355X_MICROARCH.md, inter-workgroup visibility) and so runs
// two chunks behind.  The traceback walk and the profile rebuild stay with workgroup 0, which waits for the others' "done" tokens and then
// publishes the step (new profile length, final cell) for them.  Every wait is bounded (about 3 s of s_memrealtime): a cluster that cannot
// make progress marks the interval as failed (DpMeta.pad, turned into MAUVE_ERR_HIP by the host) instead of hanging the device.
// Flag block of an interval (64-bit words, zeroed before the launch): [0, K) progress, [K, 2K) done, [2K] step done, [2K+1] profile length,
// [2K+2 .. 2K+4] the final cell, [2K+5] failure.
constexpr int DPW_KMAX = 4, DPW_FLAGS = 2 * DPW_KMAX + 8;
struct DpwCluster {
    int K, c; unsigned long long *flags; uint32_t step;
    __device__ __forceinline__ bool on() const { return K > 1; }
    __device__ __forceinline__ unsigned long long tok(int32_t q, int32_t lines) const { return ((unsigned long long)(step + 1) << 44) | ((unsigned long long)(uint32_t)q << 24) | (unsigned long long)(uint32_t)lines; }
    __device__ __forceinline__ unsigned long long *failed() const { return flags + 2 * K + 5; }
};
// by ONE wave, after its own stores: everything it has written becomes visible to a workgroup that then sees v
__device__ __forceinline__ void dpw_publish(unsigned long long *p, unsigned long long v)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// by every wave that is going to read what the token covers; false: the wait ran out (or another wave's did)
__device__ __forceinline__ bool dpw_wait(const unsigned long long *p, unsigned long long want, unsigned long long *failed)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool ok = true;
    for (;;) {
        const unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= want) break;
        if (__hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ULL) { ok = false; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > 300000000ULL) { __hip_atomic_store(failed, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
        __builtin_amdgcn_s_sleep(16);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return ok;
}

// orientation A: rows on the lanes of all waves, column by column.  Dp3A::step cut at its two exchange points.
template <int R>
struct DpwA {
    int32_t M[R], X[R], Y[R];
    int32_t s0[R], s1[R], s2[R], s3[R], gxo[R], gxe[R], E;
    int32_t Mn[R], Yn[R], Mu0, Yu0, pv; uint32_t tb;
    __device__ __forceinline__ int32_t constants(const uint32_t *Pc, int32_t i0, int32_t m, const DpScoring &sc)     // -> the lane's sum of gxe
    {
        int32_t bsum = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t cn = i0 + r < m ? Pc[i0 + r] : 0u;
            const int32_t c0 = cn & 255, c1 = (cn >> 8) & 255, c2 = (cn >> 16) & 255, c3 = cn >> 24;
            const int32_t rr = c0 + c1 + c2 + c3;
            s0[r] = c0 * sc.s[0][0] + c1 * sc.s[1][0] + c2 * sc.s[2][0] + c3 * sc.s[3][0];
            s1[r] = c0 * sc.s[0][1] + c1 * sc.s[1][1] + c2 * sc.s[2][1] + c3 * sc.s[3][1];
            s2[r] = c0 * sc.s[0][2] + c1 * sc.s[1][2] + c2 * sc.s[2][2] + c3 * sc.s[3][2];
            s3[r] = c0 * sc.s[0][3] + c1 * sc.s[1][3] + c2 * sc.s[2][3] + c3 * sc.s[3][3];
            gxo[r] = sc.go * rr; gxe[r] = sc.ge * rr; bsum += gxe[r];
            M[r] = X[r] = Y[r] = DP_NEG_INF;
        }
        return bsum;
    }
    // M and Y of column j from column j - 1.  t?o: the row above the wave's first one at column j - 1
    __device__ __forceinline__ void phase1(uint32_t b, bool j1, int32_t gyo, int32_t gye, int32_t tMo, int32_t tXo, int32_t tYo)
    {
        int32_t Md = lane0_set(wave_shr1z(M[R - 1]), tMo), Xd = lane0_set(wave_shr1z(X[R - 1]), tXo), Yd = lane0_set(wave_shr1z(Y[R - 1]), tYo);
        const bool lo = (b & 1u) != 0, hi = (b & 2u) != 0;
        tb = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int32_t bd = max(max(Md, Xd), Yd);
            const uint32_t pm = Md == bd ? 0u : (Xd == bd ? 1u : 2u);
            const int32_t sa = lo ? s1[r] : s0[r], sb = lo ? s3[r] : s2[r];
            const int32_t mv = max(bd + (hi ? sb : sa), DP_NEG_INF);
            const int32_t ya = M[r] + gyo, yb = X[r] + gyo, yc = Y[r] + gye;
            const int32_t by = max(max(ya, yb), yc);
            const uint32_t py = ya == by ? 0u : (yb == by ? 16u : 32u);
            Mn[r] = j1 ? mv : DP_NEG_INF; Yn[r] = j1 ? max(by, DP_NEG_INF) : DP_NEG_INF;
            tb |= (pm | py) << (8 * r);
            Md = M[r]; Xd = X[r]; Yd = Y[r];
        }
    }
    // the wave's own part of the scan.  tMn / tYn: the row above the wave's first one at column j
    __device__ __forceinline__ void phase2(int32_t tMn, int32_t tYn)
    {
        Mu0 = lane0_set(wave_shr1z(Mn[R - 1]), tMn); Yu0 = lane0_set(wave_shr1z(Yn[R - 1]), tYn);
        int32_t a = DP_NEG_INF, Mu = Mu0, Yu = Yu0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int32_t t = max(Mu, Yu) + gxo[r];
            a = r == 0 ? max(t, DP_NEG_INF) : max(max(t, a + gxe[r]), DP_NEG_INF);
            Mu = Mn[r]; Yu = Yn[r];
        }
        pv = wave_prefix_max(a - E);
    }
    // X of column j.  carry: max(X of the row above the super-band, aggregates of the waves above); tXn: X of the row above the wave's first one
    __device__ __forceinline__ uint32_t phase3(int32_t carry, int32_t tXn)
    {
        const int32_t xout = max(E + pv, carry + E);
        int32_t Xu = lane0_set(wave_shr1z(xout), tXn);
        int32_t Mu = Mu0, Yu = Yu0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int32_t xa = Mu + gxo[r], xb = Xu + gxe[r], xc = Yu + gxo[r];
            const int32_t bx = max(max(xa, xb), xc);
            const uint32_t px = xa == bx ? 0u : (xb == bx ? 4u : 8u);
            tb |= px << (8 * r);
            Xu = max(bx, DP_NEG_INF);
            X[r] = Xu; Mu = Mn[r]; Yu = Yn[r]; M[r] = Mn[r]; Y[r] = Yn[r];
        }
        return tb;
    }
};

// orientation B: columns on the lanes of all waves, row by row (Dp3B::step cut the same way; Y is the scanned state)
template <int R>
struct DpwB {
    int32_t M[R], X[R], Y[R];
    uint32_t bases; int32_t E;
    int32_t Mn[R], Xn[R], Ml0, Xl0, pv; uint32_t tb;
    __device__ __forceinline__ void phase1(int32_t s0, int32_t s1, int32_t s2, int32_t s3, int32_t gxo, int32_t gxe, int32_t lMo, int32_t lXo, int32_t lYo)
    {
        int32_t Md = lane0_set(wave_shr1z(M[R - 1]), lMo), Xd = lane0_set(wave_shr1z(X[R - 1]), lXo), Yd = lane0_set(wave_shr1z(Y[R - 1]), lYo);
        tb = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int32_t bd = max(max(Md, Xd), Yd);
            const uint32_t pm = Md == bd ? 0u : (Xd == bd ? 1u : 2u);
            const uint32_t b = (bases >> (2 * r)) & 3u;
            const int32_t sa = (b & 1u) ? s1 : s0, sb = (b & 1u) ? s3 : s2;
            Mn[r] = max(bd + ((b & 2u) ? sb : sa), DP_NEG_INF);
            const int32_t xa = M[r] + gxo, xb = X[r] + gxe, xc = Y[r] + gxo;
            const int32_t bx = max(max(xa, xb), xc);
            const uint32_t px = xa == bx ? 0u : (xb == bx ? 4u : 8u);
            Xn[r] = max(bx, DP_NEG_INF);
            tb |= (pm | px) << (8 * r);
            Md = M[r]; Xd = X[r]; Yd = Y[r];
        }
    }
    __device__ __forceinline__ void phase2(int32_t gyo, int32_t gye, int32_t lMn, int32_t lXn)
    {
        Ml0 = lane0_set(wave_shr1z(Mn[R - 1]), lMn); Xl0 = lane0_set(wave_shr1z(Xn[R - 1]), lXn);
        int32_t a = DP_NEG_INF, Ml = Ml0, Xl = Xl0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int32_t t = max(Ml, Xl) + gyo;
            a = r == 0 ? max(t, DP_NEG_INF) : max(max(t, a + gye), DP_NEG_INF);
            Ml = Mn[r]; Xl = Xn[r];
        }
        pv = wave_prefix_max(a - E);
    }
    __device__ __forceinline__ uint32_t phase3(int32_t gyo, int32_t gye, int32_t carry, int32_t lYn)
    {
        const int32_t yout = max(E + pv, carry + E);
        int32_t Yl = lane0_set(wave_shr1z(yout), lYn);
        int32_t Ml = Ml0, Xl = Xl0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int32_t ya = Ml + gyo, yb = Xl + gyo, yc = Yl + gye;
            const int32_t by = max(max(ya, yb), yc);
            const uint32_t py = ya == by ? 0u : (yb == by ? 16u : 32u);
            tb |= py << (8 * r);
            Yl = max(by, DP_NEG_INF);
            Y[r] = Yl; Ml = Mn[r]; Xl = Xn[r]; M[r] = Mn[r]; X[r] = Xn[r];
        }
        return tb;
    }
};

// orientation A over the whole workgroup.  Every wave of the workgroup calls it (uniform barriers); (m, n) is left in S.fin.
template <int R, int W>
__device__ __forceinline__ void dpw_sweep_a(DpwShared &S, int lane, int wv, int32_t m, int32_t n, const uint32_t *Pc, const uint8_t *seq, const DpScoring &sc,
                                            int32_t krows, int32_t *rowbuf, uint8_t *tbp, int32_t mpad, const DpwCluster &CL)
{
    constexpr int DPW_BAND = 64 * R, DPW_WAVES = W;
    static_assert(DPW_BAND * DPW_WAVES == DPW_SB, "a super-band is 2048 rows");
    const int32_t nsb = (m + DPW_SB - 1) / DPW_SB;
    const int32_t gyo = sc.go * krows, gye = sc.ge * krows;
    for (int32_t q = 0; q < nsb; q++) {
        if (CL.on() && q % CL.K != CL.c) continue;               // a cluster: this workgroup's super-bands only
        const int32_t sb0 = q * DPW_SB;
        const int32_t nw = min(DPW_WAVES, (m - sb0 + DPW_BAND - 1) / DPW_BAND);        // waves that hold rows of the profile
        const bool act = wv < nw;
        DpwA<R> L;
        const int32_t i0 = sb0 + wv * DPW_BAND + lane * R;
        const int32_t bsum = L.constants(Pc, i0, m, sc);
        const int32_t el = wave_prefix_sum(bsum);
        if (lane == 63) S.esum[wv] = el;
        __syncthreads();
        int32_t eoff = 0;
        for (int u = 0; u < wv; u++) eoff += S.esum[u];
        L.E = el + eoff;                                         // prefix sums of gxe over the rows of the SUPER-band
        // parked rows: two buffers by parity for a workgroup on its own; a cluster has several super-bands in flight: one buffer per super-band
        const int32_t *rin = rowbuf + (size_t)(CL.on() ? q - 1 : ((q & 1) ^ 1)) * 3 * (n + 1);
        int32_t *rout = rowbuf + (size_t)(CL.on() ? q : (q & 1)) * 3 * (n + 1);
        const bool park = q + 1 < nsb && wv == DPW_WAVES - 1;     // (a super-band that is followed by another one is full)
        const bool writes = act && i0 < mpad;
        const bool feed = CL.on() && q > 0;                       // the row above comes from another workgroup: wait for its progress before every fetch
        const unsigned long long *ptok = CL.flags + (feed ? (q - 1) % CL.K : 0);
        bool good = true;
        // per column from outside the super-band: the base, and the parked row of the super-band above (dp3_sweep_a's chunks; every wave
        // keeps its own copy: the base and the X of that row are needed by all of them)
        auto chunk = [&](int32_t k, uint32_t &sq, int32_t &cM, int32_t &cX, int32_t &cY) {
            const int32_t col = 64 * k + lane;
            sq = (uint32_t)seq[min(max(col - 1, 0), n - 1)];
            if (q == 0) { cM = col == 0 ? 0 : DP_NEG_INF; cX = DP_NEG_INF; cY = col == 0 ? DP_NEG_INF : gyo + (col - 1) * gye; }
            else { const int32_t cc = min(col, n); cM = rin[cc]; cX = rin[(n + 1) + cc]; cY = rin[2 * (n + 1) + cc]; }
        };
        uint32_t sq_cur, sq_nxt; int32_t cM_cur, cX_cur, cY_cur, cM_nxt, cX_nxt, cY_nxt;
        if (feed) good &= dpw_wait(ptok, CL.tok(q - 1, min(63, n) + 1), CL.failed());
        chunk(0, sq_nxt, cM_nxt, cX_nxt, cY_nxt);
        int32_t tMo = DP_NEG_INF, tXo = DP_NEG_INF, tYo = DP_NEG_INF;
        uint8_t *tw = tbp + i0;                                   // one byte per row, R of them per lane: dp3's layout for R = 4 (P(x) = x)
        for (int32_t k = 0; 64 * k <= n; k++) {
            sq_cur = sq_nxt; cM_cur = cM_nxt; cX_cur = cX_nxt; cY_cur = cY_nxt;
            if (feed && 64 * (k + 1) <= n) good &= dpw_wait(ptok, CL.tok(q - 1, min(64 * k + 127, n) + 1), CL.failed());
            chunk(k + 1, sq_nxt, cM_nxt, cX_nxt, cY_nxt);
            const int32_t jend = min(64 * k + 63, n);
            for (int32_t j = 64 * k; j <= jend; j++, tw += mpad) {
                const int sel = j & 63;
                const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int32_t)sq_cur, sel);
                const int32_t sM = __builtin_amdgcn_readlane(cM_cur, sel), sX = __builtin_amdgcn_readlane(cX_cur, sel), sY = __builtin_amdgcn_readlane(cY_cur, sel);
                if (act) {
                    L.phase1(b, j >= 1, gyo, gye, tMo, tXo, tYo);
                    if (lane == 63) { S.botM[wv] = L.Mn[R - 1]; S.botE[wv] = L.Yn[R - 1]; }
                }
                __syncthreads();
                int32_t tMn = sM, tYn = sY;
                if (act) {
                    if (wv > 0) { tMn = S.botM[wv - 1]; tYn = S.botE[wv - 1]; }
                    L.phase2(tMn, tYn);
                    if (lane == 63) S.agg[wv] = L.pv;
                }
                __syncthreads();
                if (act) {
                    const int32_t carry = max(sX, dpw_carry(S.agg, wv, lane));
                    const int32_t tXn = wv > 0 ? carry + eoff : sX;      // = what the last row of the wave above holds: E_last + max(its aggregate, its carry)
                    const uint32_t tbw = L.phase3(carry, tXn);
                    if (writes) { if (R == 4) *reinterpret_cast<uint32_t *>(tw) = tbw; else *reinterpret_cast<uint16_t *>(tw) = (uint16_t)tbw; }
                    if (park && lane == 63) { rout[j] = L.M[R - 1]; rout[(n + 1) + j] = L.X[R - 1]; rout[2 * (n + 1) + j] = L.Y[R - 1]; }
                    tMo = tMn; tXo = tXn; tYo = tYn;
                }
            }
            if (CL.on() && park) dpw_publish(CL.flags + CL.c, CL.tok(q, jend + 1));       // (wave-uniform: the parking wave) columns 0 .. jend are out
        }
        if (!good && lane == 0) __hip_atomic_store(CL.failed(), 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (q == nsb - 1) {
            const int32_t rel = m - 1 - sb0;                        // the row of (m, .) inside the super-band
            if (wv == rel / DPW_BAND && lane == (rel % DPW_BAND) / R) {
                const int ro = rel % R;
                int32_t a = L.M[0], b2 = L.X[0], c2 = L.Y[0];
#pragma unroll
                for (int r = 1; r < R; r++) if (ro == r) { a = L.M[r]; b2 = L.X[r]; c2 = L.Y[r]; }
                S.fin[0] = a; S.fin[1] = b2; S.fin[2] = c2;
            }
        }
        __threadfence_block();
        __syncthreads();                                          // parked row, esum and fin before anybody goes on
    }
}

// orientation B over the whole workgroup
template <int R, int W>
__device__ __forceinline__ void dpw_sweep_b(DpwShared &S, int lane, int wv, int32_t m, int32_t n, const uint32_t *Pc, const uint8_t *seq, const DpScoring &sc,
                                            int32_t krows, int32_t *rowbuf, uint8_t *tbp, int32_t npad, const DpwCluster &CL)
{
    constexpr int DPW_BAND = 64 * R, DPW_WAVES = W;
    static_assert(DPW_BAND * DPW_WAVES == DPW_SB, "a super-band is 2048 rows");
    const int32_t nsb = (n + DPW_SB - 1) / DPW_SB;
    const int32_t gyo = sc.go * krows, gye = sc.ge * krows;
    for (int32_t q = 0; q < nsb; q++) {
        if (CL.on() && q % CL.K != CL.c) continue;
        const int32_t sb0 = q * DPW_SB;
        const int32_t nw = min(DPW_WAVES, (n - sb0 + DPW_BAND - 1) / DPW_BAND);
        const bool act = wv < nw;
        DpwB<R> L;
        const int32_t j0 = sb0 + wv * DPW_BAND + lane * R;          // 0-based first column index (j - 1) of the lane
        L.bases = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            L.bases |= (uint32_t)(j0 + r < n ? seq[j0 + r] : 0) << (2 * r);
            L.M[r] = DP_NEG_INF; L.X[r] = DP_NEG_INF; L.Y[r] = gyo + (j0 + r) * gye;     // row 0: only Y exists (analytic, not clamped)
        }
        const int32_t eoff = wv * DPW_BAND * gye;
        L.E = (lane + 1) * R * gye + eoff;                          // prefix sums of gye over the columns of the SUPER-band
        const int32_t *cin = rowbuf + (size_t)(CL.on() ? q - 1 : ((q & 1) ^ 1)) * 3 * (m + 1);
        int32_t *cout = rowbuf + (size_t)(CL.on() ? q : (q & 1)) * 3 * (m + 1);
        const bool park = q + 1 < nsb && wv == DPW_WAVES - 1;
        const bool writes = act && j0 < npad;
        const bool feed = CL.on() && q > 0;                       // (lines of the parked column: row i is line i, row 0 included: i + 1 of them after row i)
        const unsigned long long *ptok = CL.flags + (feed ? (q - 1) % CL.K : 0);
        bool good = true;
        if (feed) good &= dpw_wait(ptok, CL.tok(q - 1, min(64, m) + 1), CL.failed());
        auto chunk = [&](int32_t k, uint32_t &pc, int32_t &cM, int32_t &cX, int32_t &cY) {
            const int32_t row = min(64 * k + lane + 1, m);
            pc = Pc[row - 1];
            if (q == 0) { cM = cX = cY = DP_NEG_INF; }
            else { cM = cin[row]; cX = cin[(m + 1) + row]; cY = cin[2 * (m + 1) + row]; }
        };
        // the column left of the wave's first one at row 0: (0, 0) for the first wave of all, the analytic row 0 otherwise (what the band to the
        // left holds there: M and X do not exist, Y is the gap from the corner); the parked column of the super-band to the left has the same
        int32_t lMo, lXo, lYo;
        if (wv == 0 && q == 0) { lMo = 0; lXo = DP_NEG_INF; lYo = DP_NEG_INF; }
        else if (wv == 0) { lMo = __builtin_amdgcn_readfirstlane(cin[0]); lXo = __builtin_amdgcn_readfirstlane(cin[(m + 1)]); lYo = __builtin_amdgcn_readfirstlane(cin[2 * (m + 1)]); }
        else { lMo = DP_NEG_INF; lXo = DP_NEG_INF; lYo = gyo + (sb0 + wv * DPW_BAND - 1) * gye; }
        // (the super-band's own left column at row i - 1, for the chain down column 0 of the first super-band)
        int32_t sMo = lMo, sXo = lXo, sYo = lYo;
        if (wv != 0) { if (q == 0) { sMo = 0; sXo = DP_NEG_INF; sYo = DP_NEG_INF; } }
        if (park && lane == 63) { cout[0] = L.M[R - 1]; cout[(m + 1)] = L.X[R - 1]; cout[2 * (m + 1)] = L.Y[R - 1]; }
        uint32_t pc_cur, pc_nxt; int32_t cM_cur, cX_cur, cY_cur, cM_nxt, cX_nxt, cY_nxt;
        chunk(0, pc_nxt, cM_nxt, cX_nxt, cY_nxt);
        uint8_t *tw = tbp + j0;
        for (int32_t k = 0; 64 * k < m; k++) {
            pc_cur = pc_nxt; cM_cur = cM_nxt; cX_cur = cX_nxt; cY_cur = cY_nxt;
            if (feed && 64 * (k + 1) < m) good &= dpw_wait(ptok, CL.tok(q - 1, min(64 * k + 128, m) + 1), CL.failed());
            chunk(k + 1, pc_nxt, cM_nxt, cX_nxt, cY_nxt);
            const int32_t iend = min(64 * k + 64, m);
            for (int32_t i = 64 * k + 1; i <= iend; i++, tw += npad) {
                const int sel = (i - 1) & 63;
                const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int32_t)pc_cur, sel);
                const int32_t c0 = c & 255, c1 = (c >> 8) & 255, c2 = (c >> 16) & 255, c3 = c >> 24, rr = c0 + c1 + c2 + c3;
                const int32_t s0 = c0 * sc.s[0][0] + c1 * sc.s[1][0] + c2 * sc.s[2][0] + c3 * sc.s[3][0];
                const int32_t s1 = c0 * sc.s[0][1] + c1 * sc.s[1][1] + c2 * sc.s[2][1] + c3 * sc.s[3][1];
                const int32_t s2 = c0 * sc.s[0][2] + c1 * sc.s[1][2] + c2 * sc.s[2][2] + c3 * sc.s[3][2];
                const int32_t s3 = c0 * sc.s[0][3] + c1 * sc.s[1][3] + c2 * sc.s[2][3] + c3 * sc.s[3][3];
                const int32_t gxo = sc.go * rr, gxe = sc.ge * rr;
                // the column left of the SUPER-band at row i (wave-uniform, every wave computes it: its Y is the carry's floor)
                int32_t sMn, sXn, sYn;
                if (q == 0) {
                    const int32_t xa = sMo + gxo, xb = sXo + gxe, xc = sYo + gxo, bx = max(max(xa, xb), xc);
                    sMn = DP_NEG_INF; sYn = DP_NEG_INF; sXn = max(bx, DP_NEG_INF);
                } else { sMn = __builtin_amdgcn_readlane(cM_cur, sel); sXn = __builtin_amdgcn_readlane(cX_cur, sel); sYn = __builtin_amdgcn_readlane(cY_cur, sel); }
                if (act) {
                    L.phase1(s0, s1, s2, s3, gxo, gxe, lMo, lXo, lYo);
                    if (lane == 63) { S.botM[wv] = L.Mn[R - 1]; S.botE[wv] = L.Xn[R - 1]; }
                }
                __syncthreads();
                int32_t lMn = sMn, lXn = sXn;
                if (act) {
                    if (wv > 0) { lMn = S.botM[wv - 1]; lXn = S.botE[wv - 1]; }
                    L.phase2(gyo, gye, lMn, lXn);
                    if (lane == 63) S.agg[wv] = L.pv;
                }
                __syncthreads();
                if (act) {
                    const int32_t carry = max(sYn, dpw_carry(S.agg, wv, lane));
                    const int32_t lYn = wv > 0 ? carry + eoff : sYn;
                    const uint32_t tbw = L.phase3(gyo, gye, carry, lYn);
                    if (writes) { if (R == 4) *reinterpret_cast<uint32_t *>(tw) = tbw; else *reinterpret_cast<uint16_t *>(tw) = (uint16_t)tbw; }
                    if (park && lane == 63) { cout[i] = L.M[R - 1]; cout[(m + 1) + i] = L.X[R - 1]; cout[2 * (m + 1) + i] = L.Y[R - 1]; }
                    lMo = lMn; lXo = lXn; lYo = lYn;
                }
                sMo = sMn; sXo = sXn; sYo = sYn;
            }
            if (CL.on() && park) dpw_publish(CL.flags + CL.c, CL.tok(q, iend + 1));       // rows 0 .. iend of the parked column are out
        }
        if (!good && lane == 0) __hip_atomic_store(CL.failed(), 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (q == nsb - 1) {
            const int32_t rel = n - 1 - sb0;
            if (wv == rel / DPW_BAND && lane == (rel % DPW_BAND) / R) {
                const int ro = rel % R;
                int32_t a = L.M[0], b2 = L.X[0], c2 = L.Y[0];
#pragma unroll
                for (int r = 1; r < R; r++) if (ro == r) { a = L.M[r]; b2 = L.X[r]; c2 = L.Y[r]; }
                S.fin[0] = a; S.fin[1] = b2; S.fin[2] = c2;
            }
        }
        __threadfence_block();
        __syncthreads();
    }
}

// one interval, all its progressive steps, by the whole workgroup (dp3_interval's structure; the walk is wave 0's, the rebuild everybody's) -- or by a
// cluster of K workgroups (CL): they all go through the steps together, the sweeps of steps with several super-bands are shared, workgroup 0 walks and
// rebuilds and tells the others the outcome of every step
template <int R, int W>
__device__ void dp_interval_wide(int nseq, int64_t iv, const uint8_t *__restrict__ codes, const int64_t *__restrict__ seq_off, DpMeta *__restrict__ meta,
                                 uint32_t *__restrict__ cntA, uint32_t *__restrict__ maskA, uint32_t *__restrict__ cntB, uint32_t *__restrict__ maskB,
                                 uint8_t *__restrict__ tb, const int64_t *__restrict__ tb_off, int32_t *__restrict__ rows, const int64_t *__restrict__ rows_off,
                                 uint8_t *__restrict__ ops, const DpScoring &sc, DpwCluster CL)
{
    __shared__ DpwShared S;
    __shared__ __attribute__((aligned(16))) uint8_t s_wwin[DP2_TB_DW * 4];       // traceback window of the walk (wave 0)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    const bool lead = !CL.on() || CL.c == 0;                 // the workgroup that walks, rebuilds and writes the result
    DpMeta mt; mt.m = 0; mt.krows = 0; mt.cur = 0; mt.pad = 0; mt.score = 0; mt.cells = 0;
    const int64_t base = seq_off[iv * nseq];
    bool good = true;
    CL.step = 0;
    // what the lead has to say after a step: the profile length and the final cell (the others poll the step token, then read them)
    auto tell = [&](int32_t len) {
        if (!CL.on()) return;
        if (threadIdx.x == 0) { CL.flags[2 * CL.K + 1] = (unsigned long long)(uint32_t)len; for (int t = 0; t < 3; t++) CL.flags[2 * CL.K + 2 + t] = (unsigned long long)(uint32_t)S.fin[t]; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave's profile stores
        __syncthreads();
        if (wv == 0) dpw_publish(CL.flags + 2 * CL.K, (unsigned long long)(CL.step + 1));
    };
    auto hear = [&](int32_t &len) {                          // by every wave of a workgroup that is not the lead
        good &= dpw_wait(CL.flags + 2 * CL.K, (unsigned long long)(CL.step + 1), CL.failed());
        len = (int32_t)(uint32_t)CL.flags[2 * CL.K + 1];
        if (threadIdx.x == 0) for (int t = 0; t < 3; t++) S.fin[t] = (int32_t)(uint32_t)CL.flags[2 * CL.K + 2 + t];
        __syncthreads();
    };
    for (int g = 0; g < nseq; g++) {
        const int64_t so = seq_off[iv * nseq + g];
        const int32_t n = (int32_t)(seq_off[iv * nseq + g + 1] - so);
        if (n == 0) continue;
        const uint8_t *seq = codes + so;
        uint32_t *Pc = (mt.cur ? cntB : cntA) + base, *Pm = (mt.cur ? maskB : maskA) + base;
        uint32_t *Qc = (mt.cur ? cntA : cntB) + base, *Qm = (mt.cur ? maskA : maskB) + base;
        if (mt.krows == 0) {
            if (lead) for (int32_t c = threadIdx.x; c < n; c += 64 * W) { Pc[c] = 1u << (8 * seq[c]); Pm[c] = 1u << g; }
            mt.m = n; mt.krows = 1;
            __threadfence_block();
            __syncthreads();
            if (CL.on()) { int32_t dummy; if (lead) { if (threadIdx.x == 0) S.fin[0] = S.fin[1] = S.fin[2] = 0; __syncthreads(); tell(n); } else hear(dummy); CL.step++; }
            continue;
        }
        const int32_t m = mt.m;
        uint8_t *tbp = tb + tb_off[iv];
        int32_t *rowbuf = rows + rows_off[iv];
        Dp3Walk Wk; Wk.win = s_wwin; Wk.cap = DP2_TB_DW * 4; Wk.tb = tbp;
        const bool wide_b = dpw_orient_b(m, n);
        const int32_t ldim = wide_b ? n : m;
        int32_t nsb = 0;                                          // super-bands of this step (0: the one-wave sweep)
        if (ldim > DP3_BAND) {
            // the lane dimension has several bands: all waves (of all the cluster's workgroups that have a super-band)
            nsb = (ldim + DPW_SB - 1) / DPW_SB;
            const int32_t pad = (int32_t)dp3_pad(ldim);
            Wk.orient_b = wide_b; Wk.R = 4; Wk.stride = pad;              // (a byte per row / column: the walk's R = 4 layout whatever R the sweep ran with)
            DpwCluster C1 = CL; if (nsb < 2) C1.K = 1;                    // one super-band: nothing to share (the lead runs it as if alone)
            if (C1.on() || lead) {
                if (!wide_b) dpw_sweep_a<R, W>(S, lane, wv, m, n, Pc, seq, sc, mt.krows, rowbuf, tbp, pad, C1);
                else dpw_sweep_b<R, W>(S, lane, wv, m, n, Pc, seq, sc, mt.krows, rowbuf, tbp, pad, C1);
            }
            if (C1.on()) {
                // the traceback bytes (and, from whoever ran the last super-band, the final cell) to the lead
                const int owner = (nsb - 1) % CL.K;
                if (CL.c == owner && threadIdx.x == 0) for (int t = 0; t < 3; t++) CL.flags[2 * CL.K + 2 + t] = (unsigned long long)(uint32_t)S.fin[t];
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (!lead) { if (wv == 0 && CL.c < nsb) dpw_publish(CL.flags + CL.K + CL.c, (unsigned long long)(CL.step + 1)); }
                else {
                    if (wv == 0) {
                        bool okw = true;
                        for (int cc = 1; cc < CL.K && cc < nsb; cc++) okw &= dpw_wait(CL.flags + CL.K + cc, (unsigned long long)(CL.step + 1), CL.failed());
                        if (!okw && lane == 0) __hip_atomic_store(CL.failed(), 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (owner != 0 && lane == 0) for (int t = 0; t < 3; t++) S.fin[t] = (int32_t)(uint32_t)CL.flags[2 * CL.K + 2 + t];
                    }
                    __syncthreads();
                    // (the other waves of the lead read the traceback only through wave 0's walk; the rebuild reads Pc / Pm, which nobody else wrote)
                }
            }
        } else if (lead) {
            // small in the dimension the cost rule picks: the one-wave sweep, by wave 0 (dp3_interval's choice of orientation and of R)
            const bool ob = dp3_orient_b(m, n);
            const int32_t ld1 = ob ? n : m, R1 = dp3_rows_per_lane(ld1), pad = (int32_t)dp3_pad(ld1);
            Wk.orient_b = ob; Wk.R = R1; Wk.stride = pad;
            if (wv == 0) {
                int32_t fM = DP_NEG_INF, fX = DP_NEG_INF, fY = DP_NEG_INF;
                if (!ob) {
                    if (R1 == 1) dp3_sweep_a<1>(lane, m, n, Pc, seq, sc, mt.krows, rowbuf, tbp, pad, fM, fX, fY);
                    else if (R1 == 2) dp3_sweep_a<2>(lane, m, n, Pc, seq, sc, mt.krows, rowbuf, tbp, pad, fM, fX, fY);
                    else dp3_sweep_a<4>(lane, m, n, Pc, seq, sc, mt.krows, rowbuf, tbp, pad, fM, fX, fY);
                } else {
                    if (R1 == 1) dp3_sweep_b<1>(lane, m, n, Pc, seq, sc, mt.krows, rowbuf, tbp, pad, fM, fX, fY);
                    else if (R1 == 2) dp3_sweep_b<2>(lane, m, n, Pc, seq, sc, mt.krows, rowbuf, tbp, pad, fM, fX, fY);
                    else dp3_sweep_b<4>(lane, m, n, Pc, seq, sc, mt.krows, rowbuf, tbp, pad, fM, fX, fY);
                }
                if (lane == 0) { S.fin[0] = fM; S.fin[1] = fX; S.fin[2] = fY; }
            }
            __threadfence_block();
            __syncthreads();
        }
        int32_t len = 0;
        if (lead) {
            const int32_t fM = S.fin[0], fX = S.fin[1], fY = S.fin[2];
            int state = 0; { int32_t best = fM; if (fX > best) { best = fX; state = 1; } if (fY > best) { best = fY; state = 2; } }
            uint8_t *opr = ops + base;                         // reversed ops, capacity m + n
            if (wv == 0) {
                const int32_t l0 = dp3_walk(Wk, m, n, state, opr, lane);
                if (lane == 0) S.len = l0;
            }
            __threadfence_block();
            __syncthreads();
            len = S.len;
            // ---- new profile: every wave keeps the running source counts, chunk k is written by wave k mod W ----
            int32_t carry_p = 0, carry_s = 0;
            for (int32_t c0i = 0, k = 0; c0i < len; c0i += 64, k++) {
                const int32_t c = c0i + lane;
                const bool ok = c < len;
                const uint32_t op = ok ? opr[len - 1 - c] : 0u;
                const uint64_t bp = __ballot(ok && (op & 1)), bs = __ballot(ok && (op & 2));
                if (ok && (k % W) == wv) {
                    const int32_t pi = carry_p + (int32_t)__popcll(bp & lt), sj = carry_s + (int32_t)__popcll(bs & lt);
                    uint32_t cv = 0, mv = 0;
                    if (op & 1) { cv = Pc[pi]; mv = Pm[pi]; }
                    if (op & 2) { cv += 1u << (8 * seq[sj]); mv |= 1u << g; }
                    Qc[c] = cv; Qm[c] = mv;
                }
                carry_p += (int32_t)__popcll(bp); carry_s += (int32_t)__popcll(bs);
            }
            tell(len);
        } else hear(len);
        {
            const int32_t fM = S.fin[0], fX = S.fin[1], fY = S.fin[2];
            int32_t best = fM; if (fX > best) best = fX; if (fY > best) best = fY;
            mt.cells += (int64_t)m * n; mt.score += best; mt.m = len; mt.krows += 1; mt.cur ^= 1;
        }
        CL.step++;
        __threadfence_block();
        __syncthreads();
    }
    if (lead && threadIdx.x == 0) {
        if (CL.on() && (!good || __hip_atomic_load(CL.failed(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ULL)) mt.pad = 1;     // a wait of the cluster ran out: the host refuses the batch
        meta[iv] = mt;
    }
}

// The workgroup launches (second stream), both over the same list entries: dp_step_wide takes the intervals the scans admit, dp_step_big --
// the systolic stripe pipeline (dp_interval_mw) -- the banded ones and whatever the scans do not admit; each leaves the
// other's entries alone (a workgroup-uniform test).  Two kernels, so that neither pays for the other's registers.
__device__ __forceinline__ bool dp_takes_wide(int nseq, int64_t iv, const int64_t *__restrict__ seq_off, const DpScoring &sc, int64_t band_from, int wide)
{
    int64_t longest = 0;
    for (int g = 0; g < nseq; g++) longest = max(longest, seq_off[iv * nseq + g + 1] - seq_off[iv * nseq + g]);
    return wide && longest <= band_from && dp3_admissible(seq_off[(iv + 1) * nseq] - seq_off[iv * nseq], nseq, sc.ge, sc.go);
}
__global__ void __launch_bounds__(64 * DP_MW_WAVES) dp_step_big(int nseq, const int64_t *__restrict__ list, const uint8_t *__restrict__ codes,
                                               const int64_t *__restrict__ seq_off, DpMeta *__restrict__ meta,
                                               uint32_t *__restrict__ cntA, uint32_t *__restrict__ maskA,
                                               uint32_t *__restrict__ cntB, uint32_t *__restrict__ maskB,
                                               uint8_t *__restrict__ tb, const int64_t *__restrict__ tb_off,
                                               int32_t *__restrict__ rows, const int64_t *__restrict__ rows_off,
                                               uint8_t *__restrict__ ops, DpScoring sc, int64_t band_from, int wide)
{
    const int64_t iv = list[blockIdx.x];
    if (dp_takes_wide(nseq, iv, seq_off, sc, band_from, wide)) return;
    dp_interval_mw(nseq, iv, codes, seq_off, meta, cntA, maskA, cntB, maskB, tb, tb_off, rows, rows_off, ops, sc, band_from);
}
template <int R, int W>
__global__ void __launch_bounds__(64 * W) dp_step_wide(int nseq, const int64_t *__restrict__ list, const uint8_t *__restrict__ codes,
                                               const int64_t *__restrict__ seq_off, DpMeta *__restrict__ meta,
                                               uint32_t *__restrict__ cntA, uint32_t *__restrict__ maskA,
                                               uint32_t *__restrict__ cntB, uint32_t *__restrict__ maskB,
                                               uint8_t *__restrict__ tb, const int64_t *__restrict__ tb_off,
                                               int32_t *__restrict__ rows, const int64_t *__restrict__ rows_off,
                                               uint8_t *__restrict__ ops, DpScoring sc, int64_t band_from, int K, unsigned long long *__restrict__ flags)
{
    // K > 1: K consecutive workgroups share list entry blockIdx.x / K (a cluster); a workgroup that can never hold a super-band of this interval -- no
    // profile and no sequence of it can be longer than all its sequences together -- leaves at once
    const uint32_t entry = K > 1 ? blockIdx.x / (uint32_t)K : blockIdx.x;
    const int64_t iv = list[entry];
    if (!dp_takes_wide(nseq, iv, seq_off, sc, band_from, 1)) return;
    DpwCluster CL; CL.K = 1; CL.c = 0; CL.flags = nullptr; CL.step = 0;
    if (K > 1) {
        const int64_t total = seq_off[(iv + 1) * nseq] - seq_off[iv * nseq];
        const int kk = (int)min((int64_t)K, (total + DPW_SB - 1) / DPW_SB);      // workgroups that can have work: the same for every workgroup of the cluster
        const int c = (int)(blockIdx.x % (uint32_t)K);
        if (c >= kk) return;
        CL.K = kk; CL.c = c; CL.flags = flags + (size_t)entry * DPW_FLAGS;
    }
    dp_interval_wide<R, W>(nseq, iv, codes, seq_off, meta, cntA, maskA, cntB, maskB, tb, tb_off, rows, rows_off, ops, sc, CL);
}

// The register-blocked launch: block ranges [one wave per interval | G = 16 | G = 8 | G = 4], the long ones first.
__global__ void __launch_bounds__(64 * DP2_WAVES) dp_step2(int nseq, const int64_t *__restrict__ list, DpClasses cl, const uint8_t *__restrict__ codes,
                                               const int64_t *__restrict__ seq_off, DpMeta *__restrict__ meta,
                                               uint32_t *__restrict__ cntA, uint32_t *__restrict__ maskA,
                                               uint32_t *__restrict__ cntB, uint32_t *__restrict__ maskB,
                                               uint8_t *__restrict__ tb, const int64_t *__restrict__ tb_off,
                                               int32_t *__restrict__ rows, const int64_t *__restrict__ rows_off,
                                               uint8_t *__restrict__ ops, DpScoring sc)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_tb[DP2_WAVES][DP2_TB_DW];
    __shared__ uint16_t s_rec[DP2_WAVES][DP2_REC];
    __shared__ uint8_t s_seq[DP2_WAVES][DP2_SEQ];
    const int wv = threadIdx.x >> 6;
    int per = 1; bool only_failed = false;
    int64_t pos0, pstep, pend;
    if (blockIdx.x >= cl.blocks_med) {
        uint32_t b = blockIdx.x - cl.blocks_med, nb; int k = 0; int64_t first, count;
        if (b < cl.blocks_c) { nb = cl.blocks_c; first = cl.first_c; count = cl.n_c; }
        else if (b < cl.blocks_c + cl.blocks_s32) { k = 1; b -= cl.blocks_c; nb = cl.blocks_s32; first = cl.first_s32; count = cl.n_s32; }
        else { k = 2; b -= cl.blocks_c + cl.blocks_s32; nb = gridDim.x - cl.blocks_med - cl.blocks_c - cl.blocks_s32; first = cl.first_s16; count = cl.n_s16; }
        const int64_t widx = (int64_t)b * DP2_WAVES + wv, nw = (int64_t)nb * DP2_WAVES;
        if (k == 0) dp2_groups<16>(nseq, list, first, count, widx, nw, codes, seq_off, meta, cntA, maskA, cntB, maskB, s_tb[wv], s_rec[wv], s_seq[wv], sc);
        else if (k == 1) dp2_groups<8>(nseq, list, first, count, widx, nw, codes, seq_off, meta, cntA, maskA, cntB, maskB, s_tb[wv], s_rec[wv], s_seq[wv], sc);
        else dp2_groups<4>(nseq, list, first, count, widx, nw, codes, seq_off, meta, cntA, maskA, cntB, maskB, s_tb[wv], s_rec[wv], s_seq[wv], sc);
        // second look at this wave's own list positions: what a group gave up is aligned below, one interval per wave
        per = k == 0 ? 4 : (k == 1 ? 8 : 16); only_failed = true;
        pos0 = first + widx * per; pstep = nw * per; pend = first + count;
        __threadfence_block();
    } else {
        pos0 = cl.first_med + (int64_t)blockIdx.x * DP2_WAVES + wv; pstep = (int64_t)cl.blocks_med * DP2_WAVES;
        pend = cl.first_med + cl.n_med;
    }
    for (int64_t lb = pos0; lb < pend; lb += pstep)
        for (int q = 0; q < per && lb + q < pend; q++) {
            const int64_t iv = list[lb + q];
            if (only_failed && meta[iv].m != -1) continue;
            // column / row scans where 32-bit prefix sums are exact (always, for real scoring schemes); else the anti-diagonal sweep
            const bool scan = cl.scan && dp3_admissible(seq_off[(iv + 1) * nseq] - seq_off[iv * nseq], nseq, sc.ge, sc.go);
            if (scan) dp3_interval(nseq, iv, codes, seq_off, meta, cntA, maskA, cntB, maskB, tb, tb_off, rows, rows_off, ops, reinterpret_cast<uint8_t *>(s_tb[wv]), sc);
            else dp2_interval(nseq, iv, codes, seq_off, meta, cntA, maskA, cntB, maskB, tb, tb_off, rows, rows_off, ops, reinterpret_cast<uint8_t *>(s_tb[wv]), sc);
        }
}

// masks of the final profiles, compacted: cols[col_off[iv] + c]
__global__ void __launch_bounds__(256) dp_gather(int nseq, int64_t n_iv, const int64_t *__restrict__ seq_off,
                                                 const DpMeta *__restrict__ meta, const uint32_t *__restrict__ maskA,
                                                 const uint32_t *__restrict__ maskB, const int64_t *__restrict__ col_off,
                                                 uint32_t *__restrict__ cols)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t iv = wave_global; iv < n_iv; iv += nwaves) {
        const DpMeta mt = meta[iv];
        const uint32_t *src = (mt.cur ? maskB : maskA) + seq_off[iv * nseq];
        uint32_t *dst = cols + col_off[iv];
        for (int32_t c = lane; c < mt.m; c += 64) dst[c] = src[c];
    }
}

// DESIGN.md S13 (refinement objective): sum-of-pairs score of every interval's columns.  One WAVE per interval: for every pair of its non-empty
// sequence slots it goes over the columns 64 at a time -- two ballots give the pair's presence masks, a lane finds its bases by prefix
// popcounts and the state of the previous column that holds one of the two (the gap rule: open for the first column of a one-sided run, extend
// for the others; columns that hold neither are skipped) by one bit scan below itself.  (One thread per (interval, pair) walking the
// columns one dependent load after the other took 4 ms for the 6.7 kb intervals of C4's root; the 28 pairs re-read the column words from L1.)
__global__ void __launch_bounds__(256) dp_sp_scores(int nseq, int64_t n_iv, const uint8_t *__restrict__ codes, const int64_t *__restrict__ seq_off,
                                                    const uint32_t *__restrict__ cols, const int64_t *__restrict__ col_off, DpScoring sc,
                                                    unsigned long long *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const uint64_t lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t iv = wave; iv < n_iv; iv += nw) {
        const int64_t c0 = col_off[iv], c1 = col_off[iv + 1];
        uint32_t have = 0;
        for (int g = 0; g < nseq; g++) if (seq_off[iv * nseq + g + 1] > seq_off[iv * nseq + g]) have |= 1u << g;
        int64_t total = 0;
        for (int a = 0; a < nseq; a++) {
            if (!(have >> a & 1u)) continue;
            for (int b = a + 1; b < nseq; b++) {
                if (!(have >> b & 1u)) continue;
                const uint8_t *sa = codes + seq_off[iv * nseq + a], *sb = codes + seq_off[iv * nseq + b];
                int64_t na = 0, nb = 0; int prev = 0;                      // bases consumed so far; 1 / 2: the last column that held one of the two held only a / only b
                for (int64_t cb = c0; cb < c1; cb += 64) {
                    const int64_t c = cb + lane;
                    const uint32_t m = c < c1 ? cols[c] : 0u;
                    const uint64_t A = __ballot(m >> a & 1u), B = __ballot(m >> b & 1u);
                    const uint64_t rel = A | B, oa = A & ~B, ob = B & ~A;
                    const bool ha = (A >> lane) & 1ULL, hb = (B >> lane) & 1ULL;
                    if (ha | hb) {
                        int pv = prev;                                     // state in front of this lane's column
                        const uint64_t below = rel & lt;
                        if (below) { const int pbit = 63 - __clzll((long long)below); pv = (oa >> pbit & 1ULL) ? 1 : ((ob >> pbit & 1ULL) ? 2 : 0); }
                        if (ha & hb) total += sc.s[sa[na + __popcll(A & lt)] & 3][sb[nb + __popcll(B & lt)] & 3];
                        else if (ha) total += pv == 1 ? sc.ge : sc.go;
                        else total += pv == 2 ? sc.ge : sc.go;
                    }
                    if (rel) { const int pbit = 63 - __clzll((long long)rel); prev = (oa >> pbit & 1ULL) ? 1 : ((ob >> pbit & 1ULL) ? 2 : 0); }
                    na += __popcll(A); nb += __popcll(B);
                }
            }
        }
        for (int o = 32; o; o >>= 1) total += __shfl_xor(total, o);
        if (lane == 0) out[iv] = (unsigned long long)total;
    }
}

// bases of the interval sequences, gathered on the device from the resident packed genomes:
// one wave per (interval, genome) descriptor; reverse descriptors are reverse-complemented.
struct DpGenomeWords { uint64_t word_off[MAUVE_MAX_SEQ]; };

__global__ void __launch_bounds__(256) dp_gather_codes(const uint64_t *__restrict__ packed, DpGenomeWords gw,
                                                       const DpSeqDesc *__restrict__ desc, const int64_t *__restrict__ seq_off,
                                                       int64_t ndesc, uint8_t *__restrict__ codes)
{
    // sixteen lanes per sequence, four sequences per wave at a time: the sequences between two anchors are a few dozen bases, and a wave that takes
    // them one by one waits for two dependent loads (descriptor, genome word) per sequence -- 30 times in a row at C3 (45 us; 4 x fewer rounds here)
    const int lane = threadIdx.x & 63, sub = lane >> 4, sl = lane & 15;
    const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t d0 = wave_global * 4; d0 < ndesc; d0 += nwaves * 4) {
        const int64_t d = d0 + sub;
        if (d >= ndesc) continue;
        const DpSeqDesc ds = desc[d];
        const uint64_t *G = packed + gw.word_off[ds.genome];
        uint8_t *out = codes + seq_off[d];
        for (int64_t i = sl; i < ds.len; i += 16) {
            const int64_t p = ds.rev ? ds.lo0 + ds.len - 1 - i : ds.lo0 + i;
            const uint32_t b = (uint32_t)(G[p >> 5] >> (2 * (p & 31))) & 3u;
            out[i] = (uint8_t)(ds.rev ? 3u - b : b);
        }
    }
}

// Which intervals get a workgroup (dp_step_big) instead of a wave: candidates (a step with several bands) whose single-wave estimate is
// at least MAUVE_DP_WIDE_MIN scan steps, at most MAUVE_DP_BIG_MAX of them.
static int dp_class_mode()
{
    static const int m = []() { const char *e = getenv("MAUVE_DP_CLASS"); return !e ? 0 : (!strcmp(e, "bound") ? 1 : (!strcmp(e, "wild") ? 2 : 0)); }();
    return m;
}
static bool dp_old_kernels() { static const bool o = getenv("MAUVE_DP_OLD") != nullptr; return o; }   // A/B switch: the systolic one-row-per-lane kernels
static int64_t dp_wide_min() { static const int64_t f = getenv("MAUVE_DP_WIDE_MIN") ? atoll(getenv("MAUVE_DP_WIDE_MIN")) : 1024; return f; }
static int64_t dp_big_max() { static const int64_t m = getenv("MAUVE_DP_BIG_MAX") ? atoll(getenv("MAUVE_DP_BIG_MAX")) : 256; return m; }
static bool dp_wide_on() { static const bool o = getenv("MAUVE_DP_NO_WIDE") == nullptr; return o; }   // A/B switch: the workgroup entries all through the systolic stripe pipeline

// the two DP launches: dp_step_big (workgroup per interval, second stream) beside dp_step (wave / sub-wave per interval),
// over the positions [a, b) of the launch list [workgroup | one wave | two per wave | four per wave]; tb_base is
// subtracted from the traceback offsets (rounds, below)
static int dp_launch_steps(mauve_ctx *ctx, int nseq, int64_t a, int64_t b, int64_t n_big, const DpClasses &full, const int64_t *d_seq_off,
                           const int64_t *d_tb_off, const int64_t *d_rows_off, const DpScoring &sc, int64_t tb_base, int64_t band_from)
{
    auto clip = [&](int64_t first, int64_t n, int64_t &f2, int64_t &n2) { f2 = std::max(first, a); n2 = std::max<int64_t>(0, std::min(first + n, b) - f2); };
    int64_t bf, bn;
    clip(0, n_big, bf, bn);
    DpClasses cl; memset(&cl, 0, sizeof cl);
    clip(full.first_med, full.n_med, cl.first_med, cl.n_med);
    clip(full.first_c, full.n_c, cl.first_c, cl.n_c);
    clip(full.first_s32, full.n_s32, cl.first_s32, cl.n_s32);
    clip(full.first_s16, full.n_s16, cl.first_s16, cl.n_s16);
    const bool oldk = dp_old_kernels();
    static const bool no_scan = getenv("MAUVE_DP_NOSCAN") != nullptr;     // A/B switch: anti-diagonal sweep for the one-wave class too
    cl.scan = no_scan ? 0 : 1;
    // intervals per workgroup: systolic kernels 4 waves x {1, 2, 4}; register-blocked kernels 2 waves x {1, 4, 8, 16}
    const int64_t wpb = oldk ? 4 : DP2_WAVES, cap = oldk ? 256 * 8 : 256 * 16;
    cl.blocks_med = (uint32_t)std::min<int64_t>((cl.n_med + wpb - 1) / wpb, cap);
    cl.blocks_c = (uint32_t)std::min<int64_t>((cl.n_c + 4 * wpb - 1) / (4 * wpb), cap);
    cl.blocks_s32 = (uint32_t)std::min<int64_t>(oldk ? (cl.n_s32 + 7) / 8 : (cl.n_s32 + 8 * wpb - 1) / (8 * wpb), cap);
    const uint32_t blocks = cl.blocks_med + cl.blocks_c + cl.blocks_s32 + (uint32_t)std::min<int64_t>(oldk ? (cl.n_s16 + 15) / 16 : (cl.n_s16 + 16 * wpb - 1) / (16 * wpb), cap);
    uint8_t *tb = ctx->dp_tb.as<uint8_t>() - tb_base;                   // only offsets >= tb_base are used in this round
    KernelTimer t(ctx, MAUVE_K_DP, b - a);
    // The workgroup-per-interval launches run beside the one-wave launch; THEY go first, on the main stream, and the one-wave launch follows on the
    // second stream behind an event: a two-wave workgroup of dp_step2 takes a quarter of a CU's LDS and registers, four of them leave room for
    // nothing else, and every slot one of them frees is refilled by the next -- a 1024-thread workgroup dispatched after them waits until that
    // whole launch has drained (measured at C5: the largest interval began 1.2 ms late, the stage took 3.0 ms instead of 1.8).  Dispatched a few
    // microseconds ahead, the large workgroups have their CUs and the small ones fill in around and behind them.
    hipStream_t small_stream = bn ? ctx->stream2 : ctx->stream;
    if (bn) {
        HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
        const bool wide = dp_wide_on();
        // 8 waves x 4 rows per lane; MAUVE_DP_WIDE_R2: 16 waves x 2 rows per lane (A/B: the same time per line -- the sweep is bound by vector issue, not by the
        // number of waves -- and the 128-register budget of a 1024-thread workgroup makes it spill)
        static const bool r4 = getenv("MAUVE_DP_WIDE_R2") == nullptr;
        // the first entries of the list (the largest intervals) get a CLUSTER of up to DPW_KMAX workgroups each (dp_interval_wide); few enough that all
        // of them are resident at once beside the rest.  MAUVE_DP_CLUSTER=0: off (A/B), =n: that many entries
        static const int64_t cl_max = getenv("MAUVE_DP_CLUSTER") ? atoll(getenv("MAUVE_DP_CLUSTER")) : 32;
        const int64_t n_cl = wide ? std::min<int64_t>(bn, cl_max) : 0;
        unsigned long long *flags = nullptr;
        if (n_cl) {
            HIPCHK(ctx, ctx->dp_wflags.ensure((size_t)n_cl * DPW_FLAGS * 8));
            HIPCHK(ctx, hipMemsetAsync(ctx->dp_wflags.p, 0, (size_t)n_cl * DPW_FLAGS * 8, ctx->stream));
            flags = ctx->dp_wflags.as<unsigned long long>();
        }
#define DPW_LAUNCH(RR, WW, first, count, K) hipLaunchKernelGGL((dp_step_wide<RR, WW>), dim3((uint32_t)((count) * (K))), dim3(64 * WW), 0, ctx->stream, nseq, \
                               ctx->dp_list.as<int64_t>() + bf + (first), ctx->dp_codes.as<uint8_t>(), d_seq_off, ctx->dp_meta.as<DpMeta>(), \
                               ctx->dp_prof_cnt.as<uint32_t>(), ctx->dp_prof_mask.as<uint32_t>(), ctx->dp_prof2_cnt.as<uint32_t>(), \
                               ctx->dp_prof2_mask.as<uint32_t>(), tb, d_tb_off, ctx->dp_rows.as<int32_t>(), \
                               d_rows_off, ctx->dp_score.as<uint8_t>(), sc, band_from, (int)(K), flags)
        if (wide && r4) { if (n_cl) DPW_LAUNCH(4, 8, 0, n_cl, DPW_KMAX); if (bn > n_cl) DPW_LAUNCH(4, 8, n_cl, bn - n_cl, 1); }
        else if (wide) { if (n_cl) DPW_LAUNCH(2, 16, 0, n_cl, DPW_KMAX); if (bn > n_cl) DPW_LAUNCH(2, 16, n_cl, bn - n_cl, 1); }
#undef DPW_LAUNCH
        // (the stripe pipeline: every entry without the wide sweep; with it, only where banded intervals or an inadmissible scoring scheme can occur)
        if (!wide || band_from != INT64_MAX || !dp3_admissible(1, nseq, sc.ge, sc.go))
            hipLaunchKernelGGL(dp_step_big, dim3((uint32_t)bn), dim3(64 * DP_MW_WAVES), 0, ctx->stream, nseq,
                               ctx->dp_list.as<int64_t>() + bf, ctx->dp_codes.as<uint8_t>(), d_seq_off, ctx->dp_meta.as<DpMeta>(),
                               ctx->dp_prof_cnt.as<uint32_t>(), ctx->dp_prof_mask.as<uint32_t>(), ctx->dp_prof2_cnt.as<uint32_t>(),
                               ctx->dp_prof2_mask.as<uint32_t>(), tb, d_tb_off, ctx->dp_rows.as<int32_t>(),
                               d_rows_off, ctx->dp_score.as<uint8_t>(), sc, band_from, wide ? 1 : 0);
    }
    if (blocks && oldk)
        hipLaunchKernelGGL(dp_step, dim3(blocks), dim3(256), 0, small_stream, nseq, ctx->dp_list.as<int64_t>(), cl,
                           ctx->dp_codes.as<uint8_t>(), d_seq_off, ctx->dp_meta.as<DpMeta>(), ctx->dp_prof_cnt.as<uint32_t>(),
                           ctx->dp_prof_mask.as<uint32_t>(), ctx->dp_prof2_cnt.as<uint32_t>(),
                           ctx->dp_prof2_mask.as<uint32_t>(), tb, d_tb_off, ctx->dp_rows.as<int32_t>(),
                           d_rows_off, ctx->dp_score.as<uint8_t>(), sc);
    else if (blocks)
        hipLaunchKernelGGL(dp_step2, dim3(blocks), dim3(64 * DP2_WAVES), 0, small_stream, nseq, ctx->dp_list.as<int64_t>(), cl,
                           ctx->dp_codes.as<uint8_t>(), d_seq_off, ctx->dp_meta.as<DpMeta>(), ctx->dp_prof_cnt.as<uint32_t>(),
                           ctx->dp_prof_mask.as<uint32_t>(), ctx->dp_prof2_cnt.as<uint32_t>(),
                           ctx->dp_prof2_mask.as<uint32_t>(), tb, d_tb_off, ctx->dp_rows.as<int32_t>(),
                           d_rows_off, ctx->dp_score.as<uint8_t>(), sc);
    if (bn) { HIPCHK(ctx, hipEventRecord(ctx->ev_join, ctx->stream2)); HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0)); }
    return MAUVE_OK;
}

// Traceback budget: the bytes are m x n per progressive step (1 B per cell), so a batch of long intervals -- a few
// hundred gaps near max_gapped_len, or profiles of many genomes -- would ask for more than any allocation gives.
// The launch list is cut into rounds whose traceback fits the budget (MAUVE_DP_TB_BUDGET bytes, default 8 GiB); the
// rounds run one after the other over the same buffer.  tb_list: cumulative traceback bytes in LIST order (n + 1).
static int64_t dp_tb_budget()
{
    static const int64_t b = getenv("MAUVE_DP_TB_BUDGET") ? atoll(getenv("MAUVE_DP_TB_BUDGET")) : (8LL << 30);
    return std::max<int64_t>(b, 1 << 16);
}
static int dp_launch_rounds(mauve_ctx *ctx, int nseq, int64_t n_iv, int64_t n_big, const DpClasses &cl, const int64_t *d_seq_off,
                            const int64_t *d_tb_off, const int64_t *d_rows_off, const DpScoring &sc, const int64_t *tb_list, int *rounds_out,
                            int64_t band_from)
{
    const int64_t budget = dp_tb_budget();
    int rounds = 0;
    for (int64_t a = 0; a < n_iv;) {
        int64_t b = a + 1;
        if (tb_list) {
            if (tb_list[a + 1] - tb_list[a] > budget) { ctx->err = "dp: one interval needs more traceback than MAUVE_DP_TB_BUDGET allows"; return MAUVE_ERR_LIMIT; }
            while (b < n_iv && tb_list[b + 1] - tb_list[a] <= budget) b++;
        } else b = n_iv;
        int rc = dp_launch_steps(ctx, nseq, a, b, n_big, cl, d_seq_off, d_tb_off, d_rows_off, sc, tb_list ? tb_list[a] : 0, band_from);
        if (rc) return rc;
        rounds++;
        a = b;
    }
    if (rounds_out) *rounds_out = rounds;
    return MAUVE_OK;
}

// shared core: seq_off is a host array; the codes are either uploaded from `codes` or gathered from `desc`
static int dp_core(mauve_ctx *ctx, int nseq, int64_t n_iv, const uint8_t *codes, const DpSeqDesc *desc,
                   const int64_t *seq_off, const mauve_scoring *scoring, uint32_t *cols, int64_t *col_off,
                   int64_t *score, int64_t *cells, int64_t *sp = nullptr)
{
    if (cells) *cells = 0;
    col_off[0] = 0;
    if (n_iv == 0) return MAUVE_OK;
    static const bool trace = getenv("MAUVE_TRACE") != nullptr;
    const double td0 = now_ms();
    const int64_t total = seq_off[n_iv * nseq];
    // per-interval scratch: traceback (worst profile length before each step) and parked rows
    mauve_ctx::DpHost &H = ctx->dph;
    std::vector<int64_t> &tb_off = H.tb_off, &rows_off = H.rows_off, &est = H.est;
    std::vector<uint8_t> &is_big = H.is_big;
    tb_off.resize((size_t)n_iv + 1); rows_off.resize((size_t)n_iv + 1);
    is_big.assign((size_t)n_iv, 0); est.assign((size_t)n_iv, 0);
    int64_t est_total = 0;
    const int64_t band_from = ctx->dp_band_from;
    static const bool no_mw = getenv("MAUVE_DP_ONE_WAVE") != nullptr;     // A/B switch for the workgroup path
    static const bool no_groups = getenv("MAUVE_DP_NO_GROUPS") != nullptr; // A/B switch for the sub-wave path
    std::vector<uint8_t> &cls = H.cls; cls.assign((size_t)n_iv, 1);
    int64_t tbt = 0, rwt = 0;
    // per-interval figures on the host helpers, then one sequential prefix
    std::vector<int64_t> &need_v = H.need, &nmax_v = H.nmax;
    need_v.resize((size_t)n_iv); nmax_v.resize((size_t)n_iv);
    ctx->pool->parallel_for(n_iv, 2048, [&](int64_t b, int64_t e) {
        for (int64_t iv = b; iv < e; iv++) {
            int64_t mmax = 0, mmin = 0, need = 0, nmax = 0, es = 0, longest = 0, rneed = 0; bool first = true; uint8_t big = 0;
            DpClassEst ce; ce.mode = dp_class_mode();
            for (int g = 0; g < nseq; g++) longest = std::max(longest, seq_off[iv * nseq + g + 1] - seq_off[iv * nseq + g]);
            const bool banded = longest > band_from;
            for (int g = 0; g < nseq; g++) {
                const int64_t n = seq_off[iv * nseq + g + 1] - seq_off[iv * nseq + g];
                ce.add(n);
                if (n == 0) continue;
                if (first) { first = false; mmax = mmin = n; continue; }
                const int64_t tbo = dp_tb_need(mmin, mmax, n, banded);  // the profile is at least as long as its longest member
                const int64_t tbn = banded ? tbo : std::max(tbo, std::max(dp2_tb_need(mmax, n), dp3_tb_need(mmax, n)));     // (any of the kernel families may run the interval)
                rneed = std::max(rneed, dp3_rows_need(mmax, n));
                need = std::max(need, tbn);
                nmax = std::max(nmax, n);
                // a step with several bands in one dimension spreads over the waves of a workgroup (wide sweep); weight: scan steps of one wave,
                // from the estimate of the profile length (at least its longest member, rarely much more)
                const int64_t e_m = std::min(mmax, mmin + mmin / 8 + 2);
                if (std::max(e_m, n) > DP3_BAND && !no_mw) big = 1;
                es += dp3_scan_steps(e_m, n);
                mmax += n; mmin = std::max(mmin, n);
            }
            if (banded && nmax) big = 2;                               // banded steps exist only in the workgroup kernel
            need_v[(size_t)iv] = need; nmax_v[(size_t)iv] = std::max(6 * (nmax + 1), rneed); est[(size_t)iv] = es; is_big[(size_t)iv] = big;
            // sub-wave classes: every profile the interval will see fits G rows, every step fits the LDS slice
            uint8_t k = 1;
            if (!no_groups && !banded) k = (uint8_t)(dp_old_kernels() ? ce.klass(DP_GRP_TMAX) : ce.klass2(DP2_R, DP2_T));
            cls[(size_t)iv] = k;
        }
    });
    for (int64_t iv = 0; iv < n_iv; iv++) {
        tb_off[iv] = tbt; rows_off[iv] = rwt;
        tbt += need_v[(size_t)iv]; rwt += nmax_v[(size_t)iv];                        // (nmax_v: parked-row entries)
        est_total += est[(size_t)iv];
    }
    tb_off[n_iv] = tbt; rows_off[n_iv] = rwt;

    DpScoring sc; sc.go = scoring->gap_open; sc.ge = scoring->gap_extend; memcpy(sc.s, scoring->matrix, sizeof sc.s);
    // longest intervals first: one wave per interval, so the tail of the launch is its longest interval
    std::vector<int64_t> &lst = H.lst; lst.resize((size_t)n_iv);
    {   // counting sort by size class (log2 of the traceback footprint), largest class first
        int64_t cnt[66] = {0};
        auto cls = [&](int64_t iv) { int64_t f = need_v[(size_t)iv]; int c = 0; while (f > 1) { f >>= 1; c++; } return 63 - c; };
        for (int64_t iv = 0; iv < n_iv; iv++) cnt[cls(iv) + 1]++;
        for (int c = 0; c < 65; c++) cnt[c + 1] += cnt[c];
        for (int64_t iv = 0; iv < n_iv; iv++) lst[(size_t)cnt[cls(iv)]++] = iv;
    }
    // workgroup-per-interval entries first (still largest first), one-wave entries after them
    // A 16-wave workgroup takes half a CU's wave slots, so it is for the tail of the launch: intervals that would keep ONE wave busy for
    // dp_wide_min() scan steps or more (the rest of the launch is over by then), the largest dp_big_max() of them.
    int64_t n_big = 0;
    {
        for (int64_t k = 0; k < n_iv; k++) {          // lst is largest first
            uint8_t &b = is_big[(size_t)lst[(size_t)k]];
            if (b == 1 && (n_big >= dp_big_max() || est[(size_t)lst[(size_t)k]] < dp_wide_min())) b = 0;
            n_big += b != 0;
        }
    }
    // list = [workgroup path | one wave | two per wave | four per wave], each class still largest first
    DpClasses cl; memset(&cl, 0, sizeof cl);
    {
        std::vector<int64_t> &tmp = H.lst2; tmp.resize((size_t)n_iv);
        int64_t cnt4[5] = {0, 0, 0, 0, 0};
        auto klass = [&](int64_t iv) { return is_big[(size_t)iv] ? 0 : (int)cls[(size_t)iv]; };
        for (int64_t k = 0; k < n_iv; k++) cnt4[klass(lst[(size_t)k])]++;
        int64_t pos[5] = {0, cnt4[0], cnt4[0] + cnt4[1], cnt4[0] + cnt4[1] + cnt4[2], cnt4[0] + cnt4[1] + cnt4[2] + cnt4[3]};
        cl.first_med = pos[1]; cl.n_med = cnt4[1]; cl.first_c = pos[2]; cl.n_c = cnt4[2]; cl.first_s32 = pos[3]; cl.n_s32 = cnt4[3]; cl.first_s16 = pos[4]; cl.n_s16 = cnt4[4];
        for (int64_t k = 0; k < n_iv; k++) { const int64_t iv = lst[(size_t)k]; tmp[(size_t)pos[klass(iv)]++] = iv; }
        lst.swap(tmp);
    }
    // traceback offsets follow the launch list, so that a round of the list uses one contiguous piece of the buffer
    std::vector<int64_t> &tb_list = H.tb_list; tb_list.resize((size_t)n_iv + 1);
    tbt = 0;
    for (int64_t k = 0; k < n_iv; k++) { const int64_t iv = lst[(size_t)k]; tb_off[iv] = tbt; tb_list[(size_t)k] = tbt; tbt += need_v[(size_t)iv]; }
    tb_list[(size_t)n_iv] = tbt; tb_off[n_iv] = tbt;
    const bool one_round = tbt <= dp_tb_budget();

    HIPCHK(ctx, ctx->dp_codes.ensure((size_t)total + 16));
    HIPCHK(ctx, ctx->dp_off.ensure((size_t)(n_iv * nseq + 1 + 3 * (n_iv + 1)) * sizeof(int64_t)));
    HIPCHK(ctx, ctx->dp_prof_cnt.ensure((size_t)(total + 1) * 4));
    HIPCHK(ctx, ctx->dp_prof_mask.ensure((size_t)(total + 1) * 4));
    HIPCHK(ctx, ctx->dp_prof2_cnt.ensure((size_t)(total + 1) * 4));
    HIPCHK(ctx, ctx->dp_prof2_mask.ensure((size_t)(total + 1) * 4));
    HIPCHK(ctx, ctx->dp_tb.ensure((size_t)std::min<int64_t>(tbt, dp_tb_budget()) + 64));
    HIPCHK(ctx, ctx->dp_rows.ensure((size_t)(rwt + 1) * 4));
    HIPCHK(ctx, ctx->dp_meta.ensure((size_t)n_iv * sizeof(DpMeta)));
    HIPCHK(ctx, ctx->dp_score.ensure((size_t)total + 16));           // reversed-ops scratch
    HIPCHK(ctx, ctx->dp_cols.ensure((size_t)(total + 1) * 4));
    int64_t *d_seq_off = ctx->dp_off.as<int64_t>();
    int64_t *d_tb_off = d_seq_off + (n_iv * nseq + 1);
    int64_t *d_rows_off = d_tb_off + (n_iv + 1);
    int64_t *d_col_off = d_rows_off + (n_iv + 1);
    // the three offset tables are adjacent on the device: one copy, from page-locked staging (a pageable source is
    // first copied by the runtime, synchronously)
    const size_t n_so = (size_t)(n_iv * nseq + 1), n_o = (size_t)n_iv + 1;
    const size_t desc_bytes = desc ? (size_t)n_iv * nseq * sizeof(DpSeqDesc) : 0;
    HIPCHK(ctx, ctx->pin_dp_in.ensure((n_so + 3 * n_o) * 8 + desc_bytes + 64));
    int64_t *pin_off = ctx->pin_dp_in.as<int64_t>();
    memcpy(pin_off, seq_off, n_so * 8);
    memcpy(pin_off + n_so, tb_off.data(), n_o * 8);
    memcpy(pin_off + n_so + n_o, rows_off.data(), n_o * 8);
    HIPCHK(ctx, hipMemcpyAsync(d_seq_off, pin_off, (n_so + 2 * n_o) * 8, hipMemcpyHostToDevice, ctx->stream));
    if (desc) {
        const int64_t nd = n_iv * nseq;
        HIPCHK(ctx, ctx->dp_desc.ensure((size_t)nd * sizeof(DpSeqDesc)));
        char *pin_desc = reinterpret_cast<char *>(pin_off + n_so + 3 * n_o);
        memcpy(pin_desc, desc, desc_bytes);
        HIPCHK(ctx, hipMemcpyAsync(ctx->dp_desc.p, pin_desc, desc_bytes, hipMemcpyHostToDevice, ctx->stream));
        DpGenomeWords gw; memset(&gw, 0, sizeof gw);
        for (int g = 0; g < ctx->nseq; g++) gw.word_off[g] = ctx->word_off[g];
        const uint32_t gb = (uint32_t)std::min<int64_t>((nd + 15) / 16, 256 * 8);
        hipLaunchKernelGGL(dp_gather_codes, dim3(gb), dim3(256), 0, ctx->stream, ctx->genomes.as<uint64_t>(), gw,
                           ctx->dp_desc.as<DpSeqDesc>(), d_seq_off, nd, ctx->dp_codes.as<uint8_t>());
    } else if (total) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->dp_codes.p, codes, (size_t)total, hipMemcpyHostToDevice, ctx->stream));
    }
    const double td1 = now_ms();
    HIPCHK(ctx, ctx->dp_list.ensure((size_t)n_iv * 8));
    int64_t *pin_list = pin_off + n_so + 2 * n_o;               // the fourth slot of the offsets block
    memcpy(pin_list, lst.data(), (size_t)n_iv * 8);
    HIPCHK(ctx, hipMemcpyAsync(ctx->dp_list.p, pin_list, (size_t)n_iv * 8, hipMemcpyHostToDevice, ctx->stream));
    int rounds = 1;
    { int rcl = dp_launch_rounds(ctx, nseq, n_iv, n_big, cl, d_seq_off, d_tb_off, d_rows_off, sc, one_round ? nullptr : tb_list.data(), &rounds, band_from); if (rcl) return rcl; }
    HIPCHK(ctx, hipGetLastError());
    if (ctx->shadow) { std::function<void()> f; f.swap(ctx->shadow); f(); }     // host work while the DP kernels run
    HIPCHK(ctx, ctx->pin_meta.ensure((size_t)n_iv * sizeof(DpMeta)));
    DpMeta *hm = ctx->pin_meta.as<DpMeta>();
    const double td2 = now_ms();
    HIPCHK(ctx, hipMemcpyAsync(hm, ctx->dp_meta.p, (size_t)n_iv * sizeof(DpMeta), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const double td3 = now_ms();
    int64_t tc = 0, ncell = 0;
    for (int64_t iv = 0; iv < n_iv; iv++) {
        if (hm[iv].pad) { ctx->err = "dp: a cluster of workgroups could not make progress on an interval (wide sweep)"; return MAUVE_ERR_HIP; }
        col_off[iv] = tc; tc += hm[iv].m; ncell += hm[iv].cells;
        if (score) score[iv] = hm[iv].score;
    }
    col_off[n_iv] = tc;
    if (cells) *cells = ncell;
    ctx->dp_last_col_off = d_col_off; ctx->dp_last_n = n_iv;
    if (tc) {
        HIPCHK(ctx, hipMemcpyAsync(d_col_off, col_off, (size_t)(n_iv + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        const uint32_t gblocks = (uint32_t)std::min<int64_t>((n_iv + 3) / 4, 256 * 8);
        hipLaunchKernelGGL(dp_gather, dim3(gblocks), dim3(256), 0, ctx->stream, nseq, n_iv, d_seq_off, ctx->dp_meta.as<DpMeta>(),
                           ctx->dp_prof_mask.as<uint32_t>(), ctx->dp_prof2_mask.as<uint32_t>(), d_col_off,
                           ctx->dp_cols.as<uint32_t>());
        HIPCHK(ctx, hipGetLastError());
        if (sp && nseq >= 2) {                  // the refinement's objective, from the columns while they are here
            HIPCHK(ctx, ctx->dp_sp.ensure((size_t)n_iv * 8 + 64));
            HIPCHK(ctx, hipMemsetAsync(ctx->dp_sp.p, 0, (size_t)n_iv * 8, ctx->stream));
            hipLaunchKernelGGL(dp_sp_scores, dim3((uint32_t)std::min<int64_t>((n_iv + 3) / 4, 256 * 16)), dim3(256), 0, ctx->stream, nseq, n_iv, ctx->dp_codes.as<uint8_t>(), d_seq_off,
                               ctx->dp_cols.as<uint32_t>(), d_col_off, sc, ctx->dp_sp.as<unsigned long long>());
            HIPCHK(ctx, hipGetLastError());
            HIPCHK(ctx, hipMemcpyAsync(ctx->pin_meta.p, ctx->dp_sp.p, (size_t)n_iv * 8, hipMemcpyDeviceToHost, ctx->stream));   // (the meta records were read above)
        }
        if (cols) HIPCHK(ctx, hipMemcpyAsync(cols, ctx->dp_cols.p, (size_t)tc * 4, hipMemcpyDeviceToHost, ctx->stream));     // (null: they stay in dp_cols for dp_fetch_picked)
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (sp && nseq >= 2) memcpy(sp, ctx->pin_meta.p, (size_t)n_iv * 8);
    }
    if (sp && (!tc || nseq < 2)) for (int64_t iv = 0; iv < n_iv; iv++) sp[iv] = 0;
    if (trace) {
        std::vector<int64_t> e2(est); std::sort(e2.begin(), e2.end(), std::greater<int64_t>());
        fprintf(stderr, "[trace] dp_core: %d round(s); %lld intervals: %lld workgroup, %lld one-wave, %lld two/wave, %lld four/wave; single-wave steps: total %lld (balanced %lld), top", rounds, (long long)n_iv,
                (long long)n_big, (long long)cl.n_med, (long long)cl.n_s32, (long long)cl.n_s16, (long long)est_total, (long long)(est_total / 3072));
        for (size_t i = 0; i < e2.size() && i < 8; i++) fprintf(stderr, " %lld", (long long)e2[i]);
        fprintf(stderr, "\n");
    }
    if (trace) fprintf(stderr, "[trace] dp_core: sizing+H2D %.3f ms, order+launch %.3f, kernel+meta D2H %.3f, gather+cols D2H %.3f\n",
                       td1 - td0, td2 - td1, td3 - td2, now_ms() - td3);
    return MAUVE_OK;
}


// ================================================================================================================
// Device front end of the whole-call path (mauve_align): from the anchor table straight to the DP launches.
// The host used to walk every inter-anchor gap three times (interval table, per-interval sizing, size-class order:
// ~1 us of pointer chasing per gap and genome, 2 ms at C3 for 0.6 ms of DP kernel).  Here the anchors go up once
// (flat int32 records in chain order) and kernels do the rest: gap of every anchor pair -> which gaps are aligned ->
// their descriptors, traceback / parked-row needs and step estimates -> offsets by tiled scans -> the launch order by
// two stable radix passes of seed_pass.hip (size class, then kernel class).  One small read-back (totals, and the gap
// code of every anchor for the assembly) sizes the buffers; the DP kernels and their result layout are unchanged.
// ================================================================================================================
namespace {
using namespace devscan;

struct DpFrontTotals {                       // device block read back once
    int64_t codes, tb, rows, est, n_dp, first_med, first_s32, first_s16, cols, cells, first_c, err;
};

__device__ __forceinline__ void dpf_gap(const int32_t *__restrict__ alen, const int32_t *__restrict__ ast, int N, uint32_t k, int g,
                                        int64_t &lo, int64_t &ln, bool &rev)
{
    const int64_t sa = ast[(size_t)k * N + g], sb = ast[(size_t)(k + 1) * N + g];
    int64_t hi;
    if (sa > 0) { lo = sa + alen[k]; hi = sb - 1; rev = false; }
    else { lo = -sb + alen[k + 1]; hi = -sa - 1; rev = true; }
    ln = hi - lo + 1; if (ln < 0) ln = 0;
}

// gap code of anchor k: -1 no gap behind it, -2 a gap that is emitted unaligned, 0 a gap for the DP (slot follows)
__global__ void __launch_bounds__(256) dpf_gap_flags(const int32_t *__restrict__ alen, const int32_t *__restrict__ ast,
                                                     const int32_t *__restrict__ alcb, uint32_t na, int N, int gapped, int64_t max_gapped,
                                                     int32_t *__restrict__ gapcode, uint32_t *__restrict__ totals_words)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k < 64) totals_words[k] = 0;                          // the front end's totals block (256 B), cleared here instead of by a memset
    if (k >= na) return;
    int32_t code = -1;
    if (k + 1 < na && alcb[k] == alcb[k + 1]) {
        int64_t tot = 0, mx = 0; int nonempty = 0;
        for (int g = 0; g < N; g++) {
            int64_t lo, ln; bool rv;
            dpf_gap(alen, ast, N, k, g, lo, ln, rv);
            tot += ln; mx = max(mx, ln); nonempty += ln > 0;
        }
        if (tot > 0) code = (gapped && nonempty >= 2 && mx <= max_gapped) ? 0 : -2;
    }
    gapcode[k] = code;
}
struct DpSlots {                             // DP gaps in chain order -> slots
    int32_t *gapcode; uint32_t na; uint32_t *anchor_of; DpFrontTotals *tot;
    __device__ uint32_t domain(int) const { return na; }
    __device__ bool flag(uint32_t k, int) const { return gapcode[k] == 0; }
    __device__ void each(uint32_t, uint32_t, bool, int) const {}
    __device__ void emit(uint32_t k, uint32_t o, int) const { gapcode[k] = (int32_t)o; anchor_of[o] = k; }
    __device__ void total(uint32_t n, int) const { tot->n_dp = n; }
};

// per DP interval: descriptors of its sequences, traceback bytes (worst profile length before each step), parked
// rows, single-wave step estimate, whether a step is long enough for the workgroup pipeline, sub-wave class
__global__ void __launch_bounds__(256) dpf_desc(const int32_t *__restrict__ alen, const int32_t *__restrict__ ast, int N,
                                                const uint32_t *__restrict__ anchor_of, const DpFrontTotals *__restrict__ tot,
                                                DpSeqDesc *__restrict__ desc, int64_t *__restrict__ need, int64_t *__restrict__ rowsn,
                                                int64_t *__restrict__ est, uint8_t *__restrict__ cand, uint8_t *__restrict__ cls,
                                                uint32_t *__restrict__ sizekey, uint32_t *__restrict__ slotval, int no_mw, int no_groups, int64_t band_from, int class_mode)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= (uint32_t)tot->n_dp) return;
    const uint32_t k = anchor_of[s];
    int64_t mmax = 0, mmin = 0, nd = 0, nmax = 0, es = 0, longest = 0, rneed = 0; bool first = true; uint8_t big = 0;
    DpClassEst ce; ce.mode = class_mode & 7;                 // bit 3: the systolic kernels' classes (MAUVE_DP_OLD)
    for (int g = 0; g < N; g++) {
        int64_t lo, n; bool rv;
        dpf_gap(alen, ast, N, k, g, lo, n, rv);
        longest = max(longest, n);
    }
    const bool banded = longest > band_from;
    for (int g = 0; g < N; g++) {
        int64_t lo, n; bool rv;
        dpf_gap(alen, ast, N, k, g, lo, n, rv);
        DpSeqDesc d; d.genome = g; d.rev = rv; d.lo0 = lo - 1; d.len = n;
        desc[(size_t)s * N + g] = d;
        ce.add(n);
        if (n == 0) continue;
        if (first) { first = false; mmax = mmin = n; continue; }
        const int64_t tbo = dp_tb_need(mmin, mmax, n, banded);
        const int64_t tbn = banded ? tbo : max(tbo, max(dp2_tb_need(mmax, n), dp3_tb_need(mmax, n)));
        rneed = max(rneed, dp3_rows_need(mmax, n));
        nd = max(nd, tbn);
        nmax = max(nmax, n);
        const int64_t e_m = min(mmax, mmin + mmin / 8 + 2);      // (dp_core: the same weights)
        if (max(e_m, n) > DP3_BAND && !no_mw) big = 1;
        es += dp3_scan_steps(e_m, n);
        mmax += n; mmin = max(mmin, n);
    }
    if (banded && nmax) big = 2;                             // banded steps exist only in the workgroup kernel
    need[s] = nd; rowsn[s] = max(6 * (nmax + 1), rneed); est[s] = es; cand[s] = big;
    uint8_t kc = 1;
    if (!no_groups && !banded) kc = (uint8_t)(class_mode & 8 ? ce.klass(DP_GRP_TMAX) : ce.klass2(DP2_R, DP2_T));
    cls[s] = kc;
    int c = 0; for (int64_t f = nd; f > 1; f >>= 1) c++;
    sizekey[s] = (uint32_t)(63 - c);                         // largest traceback footprint first
    slotval[s] = s;
}
// the same sizing for a batch whose descriptors the host made (dp_run_from_desc: the intervals of a guide-tree node and their refinement candidates)
__global__ void __launch_bounds__(256) dpf_size_desc(const DpSeqDesc *__restrict__ desc, int N, uint32_t n, DpFrontTotals *__restrict__ tot,
                                                     int64_t *__restrict__ need, int64_t *__restrict__ rowsn, int64_t *__restrict__ est, uint8_t *__restrict__ cand,
                                                     uint8_t *__restrict__ cls, uint32_t *__restrict__ sizekey, uint32_t *__restrict__ slotval, int no_mw, int no_groups,
                                                     int64_t band_from, int class_mode)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s == 0) tot->n_dp = n;                               // (the block was cleared by a memset in front of this launch)
    if (s >= n) return;
    int64_t mmax = 0, mmin = 0, nd = 0, nmax = 0, es = 0, longest = 0, rneed = 0; bool first = true; uint8_t big = 0;
    DpClassEst ce; ce.mode = class_mode & 7;
    for (int g = 0; g < N; g++) longest = max(longest, desc[(size_t)s * N + g].len);
    const bool banded = longest > band_from;
    for (int g = 0; g < N; g++) {
        const int64_t nn = desc[(size_t)s * N + g].len;
        ce.add(nn);
        if (nn == 0) continue;
        if (first) { first = false; mmax = mmin = nn; continue; }
        const int64_t tbo = dp_tb_need(mmin, mmax, nn, banded);
        const int64_t tbn = banded ? tbo : max(tbo, max(dp2_tb_need(mmax, nn), dp3_tb_need(mmax, nn)));
        rneed = max(rneed, dp3_rows_need(mmax, nn));
        nd = max(nd, tbn);
        nmax = max(nmax, nn);
        const int64_t e_m = min(mmax, mmin + mmin / 8 + 2);
        if (max(e_m, nn) > DP3_BAND && !no_mw) big = 1;
        es += dp3_scan_steps(e_m, nn);
        mmax += nn; mmin = max(mmin, nn);
    }
    if (banded && nmax) big = 2;
    need[s] = nd; rowsn[s] = max(6 * (nmax + 1), rneed); est[s] = es; cand[s] = big;
    uint8_t kc = 1;
    if (!no_groups && !banded) kc = (uint8_t)(class_mode & 8 ? ce.klass(DP_GRP_TMAX) : ce.klass2(DP2_R, DP2_T));
    cls[s] = kc;
    int c = 0; for (int64_t f = nd; f > 1; f >>= 1) c++;
    sizekey[s] = (uint32_t)(63 - c);
    slotval[s] = s;
}
struct DescLen { const DpSeqDesc *d; __device__ int64_t value(uint32_t i) const { return d[i].len; } };
struct ArrVal { const int64_t *a; __device__ int64_t value(uint32_t i) const { return a[i]; } };
struct ListVal { const int64_t *a; const uint32_t *order; __device__ int64_t value(uint32_t j) const { return a[order[j]]; } };
// traceback offsets follow the launch list: tb_off[slot at list position j] = cumulative bytes before j
__global__ void __launch_bounds__(256) dpf_tb_scatter(const int64_t *__restrict__ tb_list, const uint32_t *__restrict__ order, uint32_t n, int64_t *__restrict__ tb_off)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j < n) tb_off[order[j]] = tb_list[j];
}

// workgroup entries: candidates that would keep one wave busy for wide_min scan steps or more, the first maxn of them in size order
struct DpBigPick {
    const uint32_t *order; const uint8_t *cand, *cls; const int64_t *est; const DpFrontTotals *tot; uint32_t *key2; int64_t wide_min; uint32_t maxn;
    __device__ uint32_t domain(int) const { return (uint32_t)tot->n_dp; }
    __device__ bool flag(uint32_t j, int) const { const uint32_t s = order[j]; return cand[s] == 1 && est[s] >= wide_min; }
    __device__ void each(uint32_t j, uint32_t before, bool fl, int) const { key2[j] = ((fl && before < maxn) || cand[order[j]] == 2) ? 0u : (uint32_t)cls[order[j]]; }
    __device__ void emit(uint32_t, uint32_t, int) const {}
    __device__ void total(uint32_t, int) const {}
};
// list for the kernels (int64 slots) and the class boundaries of the class-sorted keys
__global__ void __launch_bounds__(256) dpf_list(const uint32_t *__restrict__ key2, const uint32_t *__restrict__ order, DpFrontTotals *__restrict__ tot,
                                                int64_t *__restrict__ list)
{
    const uint32_t n = (uint32_t)tot->n_dp, j = blockIdx.x * 256u + threadIdx.x;
    if (j < n) list[j] = order[j];
    if (j < 4) {                                              // first index with key >= j + 1
        uint32_t lo = 0, hi = n;
        while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if (key2[mid] >= j + 1) hi = mid; else lo = mid + 1; }
        if (j == 0) tot->first_med = lo; else if (j == 1) tot->first_c = lo; else if (j == 2) tot->first_s32 = lo; else tot->first_s16 = lo;
    }
}
struct MetaCols { const DpMeta *m; __device__ int64_t value(uint32_t i) const { return m[i].m; } };
struct MetaCells { const DpMeta *m; __device__ int64_t value(uint32_t i) const { return m[i].cells; } };
__global__ void __launch_bounds__(256) dpf_scores(const DpMeta *__restrict__ meta, uint32_t n, int64_t *__restrict__ score, DpFrontTotals *__restrict__ tot)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) { score[i] = meta[i].score; if (meta[i].pad) tot->err = 1; }       // pad: a cluster of workgroups gave up on this interval (dp_interval_wide)
}

}  // namespace

// anchors: na records in chain order (LCB by LCB, genome-0 order inside), host arrays (page-locked).  Out: gapcode[na]
// (-1 / -2 / DP slot), n_dp, the DP columns in *dcols (page-locked, grown here), dcol_off[n_dp + 1], dscore[n_dp].
// stay_on_device: the anchor arrays are device pointers and nothing but the totals comes back -- the gap codes, column
// offsets, scores and columns are left where ctx->dpf_out says, for the device assembly (assemble_dev.hip).
int dp_run_from_anchors(mauve_ctx *ctx, int N, int64_t na64, const int32_t *h_len, const int32_t *h_st, const int32_t *h_lcb, int gapped,
                        int64_t max_gapped_len, const mauve_scoring *scoring, int32_t *gapcode, int64_t *n_dp_out, int64_t *code_total_out,
                        PinnedBuf *dcols, std::vector<int64_t> &dcol_off, std::vector<int64_t> &dscore, int64_t *cells, int stay)
{
    // stay: 0 = host anchors, results to the host; 1 = device anchors used in place, results stay; 2 = host anchors, results stay
    const bool stay_on_device = stay != 0, anchors_in_place = stay == 1;
    static const bool trace = getenv("MAUVE_TRACE") != nullptr;
    static const bool no_mw = getenv("MAUVE_DP_ONE_WAVE") != nullptr, no_groups = getenv("MAUVE_DP_NO_GROUPS") != nullptr;
    const double t0 = now_ms();
    *n_dp_out = 0; *code_total_out = 0; if (cells) *cells = 0;
    dcol_off.assign(1, 0); dscore.clear();
    if (na64 < 2) { if (stay_on_device) { ctx->err = "dp: device tail needs two anchors"; return MAUVE_ERR_STATE; } for (int64_t k = 0; k < na64; k++) gapcode[k] = -1; return MAUVE_OK; }
    if (na64 >= (1LL << 31)) { ctx->err = "dp: too many anchors"; return MAUVE_ERR_LIMIT; }
    const uint32_t na = (uint32_t)na64, nb = (na + TILE - 1) / TILE, blocks = (na + 255) / 256;
    // device arrays sized by na (an upper bound of n_dp)
    const size_t w_anch = (size_t)na * (2 + (size_t)N) * 4;
    HIPCHK(ctx, ctx->dpf_anch.ensure(w_anch + (size_t)na * 4 * 2 + 64));                       // anchors, gapcode, anchor_of
    HIPCHK(ctx, ctx->dp_desc.ensure((size_t)na * N * sizeof(DpSeqDesc)));
    // work area: need / rows / est (int64), cand / cls (bytes), 5 x uint32 sort arrays, tile counts, tile sums (int64)
    auto up8 = [](size_t x) { return (x + 7) & ~(size_t)7; };
    const size_t o_cand = (size_t)na * 24, o_k = up8(o_cand + 2 * (size_t)na), o_bcnt = o_k + (size_t)na * 20, o_bsum = up8(o_bcnt + (size_t)nb * 4),
                 o_bsum2 = o_bsum + ((size_t)nb * N + 8) * 8, w_total = o_bsum2 + ((size_t)nb + 8) * 8;
    HIPCHK(ctx, ctx->dpf_work.ensure(w_total));
    HIPCHK(ctx, ctx->dp_off.ensure(((size_t)na * N + 1 + 3 * ((size_t)na + 1)) * sizeof(int64_t)));
    HIPCHK(ctx, ctx->dp_list.ensure((size_t)na * 8));
    HIPCHK(ctx, ctx->dp_meta.ensure((size_t)na * sizeof(DpMeta)));
    HIPCHK(ctx, ctx->dpf_tot.ensure(256));
    int32_t *alen = ctx->dpf_anch.as<int32_t>(), *ast = alen + na, *alcb = ast + (size_t)na * N, *d_gapcode = alcb + na;
    uint32_t *anchor_of = reinterpret_cast<uint32_t *>(d_gapcode + na);
    if (anchors_in_place) {          // the anchors are device arrays already: used where they are (no copy)
        alen = const_cast<int32_t *>(h_len); ast = const_cast<int32_t *>(h_st); alcb = const_cast<int32_t *>(h_lcb);
    }
    DpSeqDesc *desc = ctx->dp_desc.as<DpSeqDesc>();
    char *wk = ctx->dpf_work.as<char>();
    int64_t *need = reinterpret_cast<int64_t *>(wk), *rowsn = need + na, *est = rowsn + na;
    uint8_t *cand = reinterpret_cast<uint8_t *>(wk + o_cand), *cls = cand + na;
    uint32_t *k1 = reinterpret_cast<uint32_t *>(wk + o_k), *v1 = k1 + na, *k2 = v1 + na, *v2 = k2 + na, *k3 = v2 + na;
    uint32_t *bcnt = reinterpret_cast<uint32_t *>(wk + o_bcnt);
    int64_t *bsum = reinterpret_cast<int64_t *>(wk + o_bsum), *bsum2 = reinterpret_cast<int64_t *>(wk + o_bsum2);
    DpFrontTotals *tot = ctx->dpf_tot.as<DpFrontTotals>();
    int64_t *d_seq_off = ctx->dp_off.as<int64_t>();
    int64_t *d_tb_off = d_seq_off + ((size_t)na * N + 1), *d_rows_off = d_tb_off + (na + 1), *d_col_off = d_rows_off + (na + 1);
    if (!anchors_in_place) {
        HIPCHK(ctx, hipMemcpyAsync(alen, h_len, (size_t)na * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(ast, h_st, (size_t)na * N * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(alcb, h_lcb, (size_t)na * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    hipLaunchKernelGGL(dpf_gap_flags, dim3(blocks), dim3(256), 0, ctx->stream, alen, ast, alcb, na, N, gapped, max_gapped_len, d_gapcode, reinterpret_cast<uint32_t *>(tot));
    const DpSlots sl{d_gapcode, na, anchor_of, tot};
    hipLaunchKernelGGL((cmp_count<DpSlots>), dim3(nb), dim3(256), 0, ctx->stream, sl, bcnt);
    hipLaunchKernelGGL((cmp_write<DpSlots>), dim3(nb), dim3(256), 0, ctx->stream, sl, bcnt);
    hipLaunchKernelGGL(dpf_desc, dim3(blocks), dim3(256), 0, ctx->stream, alen, ast, N, anchor_of, tot, desc, need, rowsn, est, cand, cls, k1, v1,
                       (int)no_mw, (int)no_groups, ctx->dp_band_from, dp_class_mode() | (dp_old_kernels() ? 8 : 0));
    // the counts below are device values; the launches cover na (>= n_dp) entries and the kernels stop at n_dp.
    // Offsets: the value functors return 0 beyond n_dp because the arrays there are never read -- so clear them first.
    // (need / rows / est / desc of slots >= n_dp are not written: scan over exactly n_dp needs the count -> two-phase:
    //  read the small totals block back first.)
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, ctx->pin_dp_in.ensure(256 + (size_t)na * 4));
    DpFrontTotals *ht = ctx->pin_dp_in.as<DpFrontTotals>();
    int32_t *h_gapcode = reinterpret_cast<int32_t *>(ctx->pin_dp_in.as<char>() + 256);
    HIPCHK(ctx, hipMemcpyAsync(ht, tot, sizeof(DpFrontTotals), hipMemcpyDeviceToHost, ctx->stream));
    if (!stay_on_device) HIPCHK(ctx, hipMemcpyAsync(h_gapcode, d_gapcode, (size_t)na * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t n_dp = (uint32_t)ht->n_dp;
    if (!stay_on_device) memcpy(gapcode, h_gapcode, (size_t)na * 4);
    *n_dp_out = n_dp;
    mauve_ctx::DpFrontOut &fo = ctx->dpf_out;
    fo.alen = alen; fo.ast = ast; fo.alcb = alcb; fo.gapcode = d_gapcode; fo.col_off = d_col_off; fo.score = need; fo.cols = nullptr; fo.n_dp = n_dp; fo.n_cols = 0;
    const double t1 = now_ms();
    if (n_dp == 0) return MAUVE_OK;
    const uint32_t nbd = (n_dp + TILE - 1) / TILE, nbs = (n_dp * (uint32_t)N + TILE - 1) / TILE, blk_d = (n_dp + 255) / 256;
    hipLaunchKernelGGL((vscan_partial<int64_t, DescLen>), dim3(nbs), dim3(256), 0, ctx->stream, DescLen{desc}, n_dp * (uint32_t)N, bsum);
    hipLaunchKernelGGL((vscan_write<int64_t, DescLen>), dim3(nbs), dim3(256), 0, ctx->stream, DescLen{desc}, n_dp * (uint32_t)N, bsum, d_seq_off, &tot->codes);
    hipLaunchKernelGGL((vscan_partial<int64_t, ArrVal>), dim3(nbd), dim3(256), 0, ctx->stream, ArrVal{rowsn}, n_dp, bsum);
    hipLaunchKernelGGL((vscan_write<int64_t, ArrVal>), dim3(nbd), dim3(256), 0, ctx->stream, ArrVal{rowsn}, n_dp, bsum, d_rows_off, &tot->rows);
    hipLaunchKernelGGL((vscan_partial<int64_t, ArrVal>), dim3(nbd), dim3(256), 0, ctx->stream, ArrVal{est}, n_dp, bsum2);
    hipLaunchKernelGGL((vscan_write<int64_t, ArrVal>), dim3(nbd), dim3(256), 0, ctx->stream, ArrVal{est}, n_dp, bsum2, d_col_off /*scratch*/, &tot->est);
    // launch order: size class (stable), the workgroup-pipeline pick in that order, then kernel class (stable)
    uint32_t *ok = k1, *ov = v1;
    int rc = sort_pairs_u32(ctx, n_dp, 6, &ok, &ov, k2, v2, MAUVE_K_MISC);
    if (rc) return rc;
    uint32_t *fk = ok == k1 ? k2 : k1, *fv = ov == v1 ? v2 : v1;        // free pair
    const DpBigPick bp{ov, cand, cls, est, tot, fk, dp_wide_min(), (uint32_t)dp_big_max()};
    hipLaunchKernelGGL((cmp_count<DpBigPick>), dim3(nbd), dim3(256), 0, ctx->stream, bp, bcnt);
    hipLaunchKernelGGL((cmp_write<DpBigPick>), dim3(nbd), dim3(256), 0, ctx->stream, bp, bcnt);
    uint32_t *ck = fk, *cv = ov;
    rc = sort_pairs_u32(ctx, n_dp, 3, &ck, &cv, k3, fv, MAUVE_K_MISC);
    if (rc) return rc;
    hipLaunchKernelGGL(dpf_list, dim3(blk_d), dim3(256), 0, ctx->stream, ck, cv, tot, ctx->dp_list.as<int64_t>());
    // traceback offsets in list order (a round of the list then uses one contiguous piece of the buffer)
    int64_t *tb_list_dev = d_col_off;                                   // scratch until the results need it
    hipLaunchKernelGGL((vscan_partial<int64_t, ListVal>), dim3(nbd), dim3(256), 0, ctx->stream, ListVal{need, cv}, n_dp, bsum2);
    hipLaunchKernelGGL((vscan_write<int64_t, ListVal>), dim3(nbd), dim3(256), 0, ctx->stream, ListVal{need, cv}, n_dp, bsum2, tb_list_dev, &tot->tb);
    hipLaunchKernelGGL(dpf_tb_scatter, dim3(blk_d), dim3(256), 0, ctx->stream, tb_list_dev, cv, n_dp, d_tb_off);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(ht, tot, sizeof(DpFrontTotals), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t total = ht->codes, tbt = ht->tb, rwt = ht->rows;
    *code_total_out = total;
    DpClasses cl; memset(&cl, 0, sizeof cl);
    const int64_t n_big = ht->first_med;
    cl.first_med = ht->first_med; cl.n_med = ht->first_c - ht->first_med;
    cl.first_c = ht->first_c; cl.n_c = ht->first_s32 - ht->first_c;
    cl.first_s32 = ht->first_s32; cl.n_s32 = ht->first_s16 - ht->first_s32;
    cl.first_s16 = ht->first_s16; cl.n_s16 = (int64_t)n_dp - ht->first_s16;          // (block counts: dp_launch_steps)
    const double t2 = now_ms();
    HIPCHK(ctx, ctx->dp_codes.ensure((size_t)total + 16));
    HIPCHK(ctx, ctx->dp_prof_cnt.ensure((size_t)(total + 1) * 4));
    HIPCHK(ctx, ctx->dp_prof_mask.ensure((size_t)(total + 1) * 4));
    HIPCHK(ctx, ctx->dp_prof2_cnt.ensure((size_t)(total + 1) * 4));
    HIPCHK(ctx, ctx->dp_prof2_mask.ensure((size_t)(total + 1) * 4));
    HIPCHK(ctx, ctx->dp_tb.ensure((size_t)std::min<int64_t>(tbt, dp_tb_budget()) + 64));
    HIPCHK(ctx, ctx->dp_rows.ensure((size_t)(rwt + 1) * 4));
    HIPCHK(ctx, ctx->dp_score.ensure((size_t)total + 16));           // reversed-ops scratch
    HIPCHK(ctx, ctx->dp_cols.ensure((size_t)(total + 1) * 4));
    std::vector<int64_t> &tb_list = ctx->dph.tb_list; tb_list.clear();
    if (tbt > dp_tb_budget()) {                                         // several rounds: the host needs the cumulative bytes to cut them
        tb_list.resize((size_t)n_dp + 1);
        HIPCHK(ctx, hipMemcpy(tb_list.data(), tb_list_dev, ((size_t)n_dp + 1) * 8, hipMemcpyDeviceToHost));
    }
    {
        DpGenomeWords gw; memset(&gw, 0, sizeof gw);
        for (int g = 0; g < ctx->nseq; g++) gw.word_off[g] = ctx->word_off[g];
        const int64_t nd = (int64_t)n_dp * N;
        const uint32_t gb = (uint32_t)std::min<int64_t>((nd + 15) / 16, 256 * 8);
        hipLaunchKernelGGL(dp_gather_codes, dim3(gb), dim3(256), 0, ctx->stream, ctx->genomes.as<uint64_t>(), gw, desc, d_seq_off, nd,
                           ctx->dp_codes.as<uint8_t>());
    }
    DpScoring sc; sc.go = scoring->gap_open; sc.ge = scoring->gap_extend; memcpy(sc.s, scoring->matrix, sizeof sc.s);
    int rounds = 1;
    rc = dp_launch_rounds(ctx, N, n_dp, n_big, cl, d_seq_off, d_tb_off, d_rows_off, sc, tb_list.empty() ? nullptr : tb_list.data(), &rounds, ctx->dp_band_from);
    if (rc) return rc;
    // results: column offsets, scores and the cell count by scans over the per-interval records; the columns compacted
    const DpMeta *meta = ctx->dp_meta.as<DpMeta>();
    int64_t *d_score = need;                                            // the sizing arrays are free again
    hipLaunchKernelGGL((vscan_partial<int64_t, MetaCols>), dim3(nbd), dim3(256), 0, ctx->stream, MetaCols{meta}, n_dp, bsum);
    hipLaunchKernelGGL((vscan_write<int64_t, MetaCols>), dim3(nbd), dim3(256), 0, ctx->stream, MetaCols{meta}, n_dp, bsum, d_col_off, &tot->cols);
    hipLaunchKernelGGL((vscan_partial<int64_t, MetaCells>), dim3(nbd), dim3(256), 0, ctx->stream, MetaCells{meta}, n_dp, bsum2);
    hipLaunchKernelGGL((vscan_write<int64_t, MetaCells>), dim3(nbd), dim3(256), 0, ctx->stream, MetaCells{meta}, n_dp, bsum2, rowsn /*scratch*/, &tot->cells);
    hipLaunchKernelGGL(dpf_scores, dim3(blk_d), dim3(256), 0, ctx->stream, meta, n_dp, d_score, tot);
    {
        const uint32_t gblocks = (uint32_t)std::min<int64_t>(((int64_t)n_dp + 3) / 4, 256 * 8);
        hipLaunchKernelGGL(dp_gather, dim3(gblocks), dim3(256), 0, ctx->stream, N, (int64_t)n_dp, d_seq_off, meta, ctx->dp_prof_mask.as<uint32_t>(),
                           ctx->dp_prof2_mask.as<uint32_t>(), d_col_off, ctx->dp_cols.as<uint32_t>());
    }
    HIPCHK(ctx, hipGetLastError());
    if (ctx->shadow) { std::function<void()> f; f.swap(ctx->shadow); f(); }     // host work while the DP kernels run
    if (stay_on_device) {
        HIPCHK(ctx, hipMemcpyAsync(ht, tot, sizeof(DpFrontTotals), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (ht->err) { ctx->err = "dp: a cluster of workgroups could not make progress on an interval (wide sweep)"; return MAUVE_ERR_HIP; }
        if (cells) *cells = ht->cells;
        fo.cols = ctx->dp_cols.as<uint32_t>(); fo.n_cols = ht->cols;
        if (trace) fprintf(stderr, "[trace] dp (device front, results stay): %u intervals (%lld workgroup, %lld one-wave, %lld two/wave, %lld four/wave), %d round(s); gaps+slots %.3f ms, sizing+order %.3f, kernels+offsets %.3f\n",
                           n_dp, (long long)n_big, (long long)cl.n_med, (long long)cl.n_s32, (long long)cl.n_s16, rounds, t1 - t0, t2 - t1, now_ms() - t2);
        return MAUVE_OK;
    }
    dcol_off.resize((size_t)n_dp + 1); dscore.resize((size_t)n_dp);
    HIPCHK(ctx, ctx->pin_meta.ensure(((size_t)n_dp * 2 + 2) * 8));
    int64_t *p_off = ctx->pin_meta.as<int64_t>(), *p_score = p_off + n_dp + 1;
    HIPCHK(ctx, hipMemcpyAsync(ht, tot, sizeof(DpFrontTotals), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(p_off, d_col_off, ((size_t)n_dp + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(p_score, d_score, (size_t)n_dp * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const double t3 = now_ms();
    if (ht->err) { ctx->err = "dp: a cluster of workgroups could not make progress on an interval (wide sweep)"; return MAUVE_ERR_HIP; }
    const int64_t tc = ht->cols;
    if (cells) *cells = ht->cells;
    memcpy(dcol_off.data(), p_off, ((size_t)n_dp + 1) * 8);
    memcpy(dscore.data(), p_score, (size_t)n_dp * 8);
    HIPCHK(ctx, dcols->ensure(((size_t)tc + 1) * 4));
    if (tc) {
        HIPCHK(ctx, hipMemcpyAsync(dcols->p, ctx->dp_cols.p, (size_t)tc * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (trace) fprintf(stderr, "[trace] dp (device front): %u intervals (%lld workgroup, %lld one-wave, %lld two/wave, %lld four/wave), %d round(s); gaps+slots %.3f ms, sizing+order %.3f, kernels+offsets %.3f, columns %.3f\n",
                       n_dp, (long long)n_big, (long long)cl.n_med, (long long)cl.n_s32, (long long)cl.n_s16, rounds, t1 - t0, t2 - t1, t3 - t2, now_ms() - t3);
    return MAUVE_OK;
}

// A batch of intervals given by descriptors (the progressive path: the intervals of a guide-tree node and all their refinement candidates, 86 000 at
// C4's root), through the device front end: the descriptors go up once, sizing, offsets and launch order are made by the kernels and scans of
// dp_run_from_anchors (dp_core's host loops took 7 ms for that batch, beside 4 ms of DP kernels), the result offsets come from scans as well.
// cols == nullptr: the columns stay in dp_cols (dp_fetch_picked).  sp: the refinement objective (dp_sp_scores), or nullptr.
// The refinement candidates of a guide-tree node made where they are used: rows [n_orig, n_orig + cbase[n_orig]) of the descriptor table are
// rotations of the intervals in front of them -- candidate cbase[iv] + r - 1 holds interval iv's non-empty sequences rotated by r, empty slots behind -- so
// the host uploads the intervals and one offset each instead of building and copying three times as many rows (C4's root: 22 MB, 2 ms of the host).
__global__ void __launch_bounds__(256) dpf_rotate_desc(DpSeqDesc *__restrict__ desc, int N, uint32_t n_orig, const int32_t *__restrict__ cbase)
{
    const uint32_t iv = blockIdx.x * 256u + threadIdx.x;
    if (iv >= n_orig) return;
    const int32_t c0 = cbase[iv], cnt = cbase[iv + 1] - c0;
    if (cnt <= 0) return;
    const DpSeqDesc *di = desc + (size_t)iv * N;
    int k = 0;
    for (int j = 0; j < N; j++) k += di[j].len != 0;
    DpSeqDesc none = di[0]; none.rev = 0; none.lo0 = 0; none.len = 0;
    for (int r = 1; r <= cnt; r++) {
        DpSeqDesc *o = desc + ((size_t)n_orig + (size_t)c0 + (size_t)(r - 1)) * N;
        // slot j takes the ((j + r) mod k)-th non-empty sequence: walk the non-empty ones once, starting at the r-th
        int t = 0, filled = 0;
        for (int j = 0; j < N; j++) {
            if (!di[j].len) continue;
            const int slot = ((t - r) % k + k) % k;            // the t-th non-empty sequence lands in slot (t - r) mod k
            o[slot] = di[j]; t++; filled++;
        }
        for (int j = filled; j < N; j++) o[j] = none;
    }
}

int dp_run_from_desc(mauve_ctx *ctx, int N, int64_t n_iv, const DpSeqDesc *h_desc, const mauve_scoring *scoring, uint32_t *cols, int64_t *col_off, int64_t *score,
                     int64_t *cells, int64_t *sp, int64_t n_orig = -1, const int32_t *cbase = nullptr)
{
    static const bool trace = getenv("MAUVE_TRACE") != nullptr;
    static const bool no_mw = getenv("MAUVE_DP_ONE_WAVE") != nullptr, no_groups = getenv("MAUVE_DP_NO_GROUPS") != nullptr;
    const double t0 = now_ms();
    if (cells) *cells = 0;
    col_off[0] = 0;
    if (n_iv == 0) return MAUVE_OK;
    if (n_iv >= (1LL << 31) / std::max(N, 1)) { ctx->err = "dp: too many intervals"; return MAUVE_ERR_LIMIT; }
    const uint32_t na = (uint32_t)n_iv, n_dp = na, nb = (na + TILE - 1) / TILE;
    HIPCHK(ctx, ctx->dp_desc.ensure((size_t)na * N * sizeof(DpSeqDesc)));
    auto up8 = [](size_t x) { return (x + 7) & ~(size_t)7; };
    const size_t o_cand = (size_t)na * 24, o_k = up8(o_cand + 2 * (size_t)na), o_bcnt = o_k + (size_t)na * 20, o_bsum = up8(o_bcnt + (size_t)nb * 4),
                 o_bsum2 = o_bsum + ((size_t)nb * N + 8) * 8, w_total = o_bsum2 + ((size_t)nb + 8) * 8;
    HIPCHK(ctx, ctx->dpf_work.ensure(w_total));
    HIPCHK(ctx, ctx->dp_off.ensure(((size_t)na * N + 1 + 3 * ((size_t)na + 1)) * sizeof(int64_t)));
    HIPCHK(ctx, ctx->dp_list.ensure((size_t)na * 8));
    HIPCHK(ctx, ctx->dp_meta.ensure((size_t)na * sizeof(DpMeta)));
    HIPCHK(ctx, ctx->dpf_tot.ensure(256));
    DpSeqDesc *desc = ctx->dp_desc.as<DpSeqDesc>();
    char *wk = ctx->dpf_work.as<char>();
    int64_t *need = reinterpret_cast<int64_t *>(wk), *rowsn = need + na, *est = rowsn + na;
    uint8_t *cand = reinterpret_cast<uint8_t *>(wk + o_cand), *cls = cand + na;
    uint32_t *k1 = reinterpret_cast<uint32_t *>(wk + o_k), *v1 = k1 + na, *k2 = v1 + na, *v2 = k2 + na, *k3 = v2 + na;
    uint32_t *bcnt = reinterpret_cast<uint32_t *>(wk + o_bcnt);
    int64_t *bsum = reinterpret_cast<int64_t *>(wk + o_bsum), *bsum2 = reinterpret_cast<int64_t *>(wk + o_bsum2);
    DpFrontTotals *tot = ctx->dpf_tot.as<DpFrontTotals>();
    int64_t *d_seq_off = ctx->dp_off.as<int64_t>();
    int64_t *d_tb_off = d_seq_off + ((size_t)na * N + 1), *d_rows_off = d_tb_off + (na + 1), *d_col_off = d_rows_off + (na + 1);
    // cbase: only the first n_orig rows come from the host, the rest are their rotations (dpf_rotate_desc)
    const bool rot = cbase != nullptr && n_orig >= 0 && n_orig < n_iv;
    const size_t rows_up = rot ? (size_t)n_orig : (size_t)na;
    const size_t desc_bytes = rows_up * N * sizeof(DpSeqDesc), cb_bytes = rot ? ((size_t)n_orig + 1) * 4 : 0;
    HIPCHK(ctx, ctx->pin_dp_in.ensure(256 + desc_bytes + cb_bytes + 64));
    DpFrontTotals *ht = ctx->pin_dp_in.as<DpFrontTotals>();
    memcpy(ctx->pin_dp_in.as<char>() + 256, h_desc, desc_bytes);
    HIPCHK(ctx, hipMemcpyAsync(desc, ctx->pin_dp_in.as<char>() + 256, desc_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (rot) {
        if ((int64_t)cbase[n_orig] != n_iv - n_orig) { ctx->err = "dp: the rotation table does not match the batch"; return MAUVE_ERR_ARG; }
        char *pcb = ctx->pin_dp_in.as<char>() + 256 + ((desc_bytes + 7) & ~(size_t)7);
        memcpy(pcb, cbase, cb_bytes);
        int32_t *d_cb = reinterpret_cast<int32_t *>(k1);                 // (the sort's key array is free until dpf_size_desc fills it)
        HIPCHK(ctx, hipMemcpyAsync(d_cb, pcb, cb_bytes, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(dpf_rotate_desc, dim3((uint32_t)((n_orig + 255) / 256)), dim3(256), 0, ctx->stream, desc, N, (uint32_t)n_orig, d_cb);
    }
    HIPCHK(ctx, hipMemsetAsync(tot, 0, 256, ctx->stream));
    const uint32_t blk_d = (n_dp + 255) / 256;
    hipLaunchKernelGGL(dpf_size_desc, dim3(blk_d), dim3(256), 0, ctx->stream, desc, N, n_dp, tot, need, rowsn, est, cand, cls, k1, v1, (int)no_mw, (int)no_groups,
                       ctx->dp_band_from, dp_class_mode() | (dp_old_kernels() ? 8 : 0));
    const uint32_t nbd = (n_dp + TILE - 1) / TILE, nbs = (n_dp * (uint32_t)N + TILE - 1) / TILE;
    hipLaunchKernelGGL((vscan_partial<int64_t, DescLen>), dim3(nbs), dim3(256), 0, ctx->stream, DescLen{desc}, n_dp * (uint32_t)N, bsum);
    hipLaunchKernelGGL((vscan_write<int64_t, DescLen>), dim3(nbs), dim3(256), 0, ctx->stream, DescLen{desc}, n_dp * (uint32_t)N, bsum, d_seq_off, &tot->codes);
    hipLaunchKernelGGL((vscan_partial<int64_t, ArrVal>), dim3(nbd), dim3(256), 0, ctx->stream, ArrVal{rowsn}, n_dp, bsum);
    hipLaunchKernelGGL((vscan_write<int64_t, ArrVal>), dim3(nbd), dim3(256), 0, ctx->stream, ArrVal{rowsn}, n_dp, bsum, d_rows_off, &tot->rows);
    hipLaunchKernelGGL((vscan_partial<int64_t, ArrVal>), dim3(nbd), dim3(256), 0, ctx->stream, ArrVal{est}, n_dp, bsum2);
    hipLaunchKernelGGL((vscan_write<int64_t, ArrVal>), dim3(nbd), dim3(256), 0, ctx->stream, ArrVal{est}, n_dp, bsum2, d_col_off /*scratch*/, &tot->est);
    uint32_t *ok = k1, *ov = v1;
    int rc = sort_pairs_u32(ctx, n_dp, 6, &ok, &ov, k2, v2, MAUVE_K_MISC);
    if (rc) return rc;
    uint32_t *fk = ok == k1 ? k2 : k1, *fv = ov == v1 ? v2 : v1;
    const DpBigPick bp{ov, cand, cls, est, tot, fk, dp_wide_min(), (uint32_t)dp_big_max()};
    hipLaunchKernelGGL((cmp_count<DpBigPick>), dim3(nbd), dim3(256), 0, ctx->stream, bp, bcnt);
    hipLaunchKernelGGL((cmp_write<DpBigPick>), dim3(nbd), dim3(256), 0, ctx->stream, bp, bcnt);
    uint32_t *ck = fk, *cv = ov;
    rc = sort_pairs_u32(ctx, n_dp, 3, &ck, &cv, k3, fv, MAUVE_K_MISC);
    if (rc) return rc;
    hipLaunchKernelGGL(dpf_list, dim3(blk_d), dim3(256), 0, ctx->stream, ck, cv, tot, ctx->dp_list.as<int64_t>());
    int64_t *tb_list_dev = d_col_off;
    hipLaunchKernelGGL((vscan_partial<int64_t, ListVal>), dim3(nbd), dim3(256), 0, ctx->stream, ListVal{need, cv}, n_dp, bsum2);
    hipLaunchKernelGGL((vscan_write<int64_t, ListVal>), dim3(nbd), dim3(256), 0, ctx->stream, ListVal{need, cv}, n_dp, bsum2, tb_list_dev, &tot->tb);
    hipLaunchKernelGGL(dpf_tb_scatter, dim3(blk_d), dim3(256), 0, ctx->stream, tb_list_dev, cv, n_dp, d_tb_off);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(ht, tot, sizeof(DpFrontTotals), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t total = ht->codes, tbt = ht->tb, rwt = ht->rows;
    DpClasses cl; memset(&cl, 0, sizeof cl);
    const int64_t n_big = ht->first_med;
    cl.first_med = ht->first_med; cl.n_med = ht->first_c - ht->first_med;
    cl.first_c = ht->first_c; cl.n_c = ht->first_s32 - ht->first_c;
    cl.first_s32 = ht->first_s32; cl.n_s32 = ht->first_s16 - ht->first_s32;
    cl.first_s16 = ht->first_s16; cl.n_s16 = (int64_t)n_dp - ht->first_s16;
    const double t1 = now_ms();
    HIPCHK(ctx, ctx->dp_codes.ensure((size_t)total + 16));
    HIPCHK(ctx, ctx->dp_prof_cnt.ensure((size_t)(total + 1) * 4));
    HIPCHK(ctx, ctx->dp_prof_mask.ensure((size_t)(total + 1) * 4));
    HIPCHK(ctx, ctx->dp_prof2_cnt.ensure((size_t)(total + 1) * 4));
    HIPCHK(ctx, ctx->dp_prof2_mask.ensure((size_t)(total + 1) * 4));
    HIPCHK(ctx, ctx->dp_tb.ensure((size_t)std::min<int64_t>(tbt, dp_tb_budget()) + 64));
    HIPCHK(ctx, ctx->dp_rows.ensure((size_t)(rwt + 1) * 4));
    HIPCHK(ctx, ctx->dp_score.ensure((size_t)total + 16));
    HIPCHK(ctx, ctx->dp_cols.ensure((size_t)(total + 1) * 4));
    std::vector<int64_t> &tb_list = ctx->dph.tb_list; tb_list.clear();
    if (tbt > dp_tb_budget()) {
        tb_list.resize((size_t)n_dp + 1);
        HIPCHK(ctx, hipMemcpy(tb_list.data(), tb_list_dev, ((size_t)n_dp + 1) * 8, hipMemcpyDeviceToHost));
    }
    {
        DpGenomeWords gw; memset(&gw, 0, sizeof gw);
        for (int g = 0; g < ctx->nseq; g++) gw.word_off[g] = ctx->word_off[g];
        const int64_t nd = (int64_t)n_dp * N;
        hipLaunchKernelGGL(dp_gather_codes, dim3((uint32_t)std::min<int64_t>((nd + 15) / 16, 256 * 8)), dim3(256), 0, ctx->stream, ctx->genomes.as<uint64_t>(), gw, desc, d_seq_off, nd,
                           ctx->dp_codes.as<uint8_t>());
    }
    DpScoring sc; sc.go = scoring->gap_open; sc.ge = scoring->gap_extend; memcpy(sc.s, scoring->matrix, sizeof sc.s);
    int rounds = 1;
    rc = dp_launch_rounds(ctx, N, n_dp, n_big, cl, d_seq_off, d_tb_off, d_rows_off, sc, tb_list.empty() ? nullptr : tb_list.data(), &rounds, ctx->dp_band_from);
    if (rc) return rc;
    const DpMeta *meta = ctx->dp_meta.as<DpMeta>();
    int64_t *d_score = need;
    hipLaunchKernelGGL((vscan_partial<int64_t, MetaCols>), dim3(nbd), dim3(256), 0, ctx->stream, MetaCols{meta}, n_dp, bsum);
    hipLaunchKernelGGL((vscan_write<int64_t, MetaCols>), dim3(nbd), dim3(256), 0, ctx->stream, MetaCols{meta}, n_dp, bsum, d_col_off, &tot->cols);
    hipLaunchKernelGGL((vscan_partial<int64_t, MetaCells>), dim3(nbd), dim3(256), 0, ctx->stream, MetaCells{meta}, n_dp, bsum2);
    hipLaunchKernelGGL((vscan_write<int64_t, MetaCells>), dim3(nbd), dim3(256), 0, ctx->stream, MetaCells{meta}, n_dp, bsum2, rowsn /*scratch*/, &tot->cells);
    hipLaunchKernelGGL(dpf_scores, dim3(blk_d), dim3(256), 0, ctx->stream, meta, n_dp, d_score, tot);
    hipLaunchKernelGGL(dp_gather, dim3((uint32_t)std::min<int64_t>(((int64_t)n_dp + 3) / 4, 256 * 8)), dim3(256), 0, ctx->stream, N, (int64_t)n_dp, d_seq_off, meta,
                       ctx->dp_prof_mask.as<uint32_t>(), ctx->dp_prof2_mask.as<uint32_t>(), d_col_off, ctx->dp_cols.as<uint32_t>());
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, ctx->pin_meta.ensure(((size_t)n_dp * 3 + 2) * 8));
    int64_t *p_off = ctx->pin_meta.as<int64_t>(), *p_score = p_off + n_dp + 1, *p_sp = p_score + n_dp;
    if (sp && N >= 2) {                     // the refinement's objective, from the columns while they are here
        HIPCHK(ctx, ctx->dp_sp.ensure((size_t)n_dp * 8 + 64));
        hipLaunchKernelGGL(dp_sp_scores, dim3((uint32_t)std::min<int64_t>(((int64_t)n_dp + 3) / 4, 256 * 16)), dim3(256), 0, ctx->stream, N, (int64_t)n_dp, ctx->dp_codes.as<uint8_t>(), d_seq_off,
                           ctx->dp_cols.as<uint32_t>(), d_col_off, sc, ctx->dp_sp.as<unsigned long long>());
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(p_sp, ctx->dp_sp.p, (size_t)n_dp * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipMemcpyAsync(ht, tot, sizeof(DpFrontTotals), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(p_off, d_col_off, ((size_t)n_dp + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(p_score, d_score, (size_t)n_dp * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ht->err) { ctx->err = "dp: a cluster of workgroups could not make progress on an interval (wide sweep)"; return MAUVE_ERR_HIP; }
    const int64_t tc = ht->cols;
    if (cells) *cells = ht->cells;
    memcpy(col_off, p_off, ((size_t)n_dp + 1) * 8);
    if (score) memcpy(score, p_score, (size_t)n_dp * 8);
    if (sp) { if (N >= 2) memcpy(sp, p_sp, (size_t)n_dp * 8); else memset(sp, 0, (size_t)n_dp * 8); }
    ctx->dp_last_col_off = d_col_off; ctx->dp_last_n = n_dp;
    if (cols && tc) {
        if (host_pointer_is_pinned(cols)) HIPCHK(ctx, hipMemcpyAsync(cols, ctx->dp_cols.p, (size_t)tc * 4, hipMemcpyDeviceToHost, ctx->stream));
        else HIPCHK(ctx, hipMemcpy(cols, ctx->dp_cols.p, (size_t)tc * 4, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (trace) fprintf(stderr, "[trace] dp (descriptor batch, device front): %u intervals (%lld workgroup, %lld one-wave, %lld two/wave, %lld four/wave), %d round(s); upload+sizing+order %.3f ms, kernels+results %.3f\n",
                       n_dp, (long long)n_big, (long long)cl.n_med, (long long)cl.n_s32, (long long)cl.n_s16, rounds, t1 - t0, now_ms() - t1);
    return MAUVE_OK;
}

// columns of picked intervals of the last dp_core batch, one behind the other: a wave per pick
__global__ void __launch_bounds__(256) dp_pick_cols(const uint32_t *__restrict__ cols, const int64_t *__restrict__ col_off, const int64_t *__restrict__ pick,
                                                    const int64_t *__restrict__ out_off, int64_t n_pick, uint32_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t q = wave; q < n_pick; q += nw) {
        const int64_t iv = pick[q], a = col_off[iv], n = col_off[iv + 1] - a, o = out_off[q];
        for (int64_t c = lane; c < n; c += 64) out[o + c] = cols[a + c];
    }
}

// The columns of n_pick intervals of the batch dp_core has just run with cols == nullptr (their ids, ascending or not), compacted on the device and
// copied out in pick order; col_off: the batch's host offsets (lengths), out_off[n_pick + 1]: where each pick's columns start in `out`.
int dp_fetch_picked(mauve_ctx *ctx, int64_t n_pick, const int64_t *pick, const int64_t *col_off, uint32_t *out, int64_t *out_off)
{
    out_off[0] = 0;
    for (int64_t q = 0; q < n_pick; q++) {
        if (pick[q] < 0 || pick[q] >= ctx->dp_last_n) { ctx->err = "dp_fetch_picked: interval outside the last batch"; return MAUVE_ERR_ARG; }
        out_off[q + 1] = out_off[q] + (col_off[pick[q] + 1] - col_off[pick[q]]);
    }
    const int64_t total = out_off[n_pick];
    if (!total) return MAUVE_OK;
    HIPCHK(ctx, ctx->dp_pick.ensure((size_t)(2 * n_pick + 1) * 8 + (size_t)total * 4 + 64));
    HIPCHK(ctx, ctx->pin_dp_in.ensure((size_t)(2 * n_pick + 1) * 8));
    int64_t *hp = ctx->pin_dp_in.as<int64_t>();
    memcpy(hp, pick, (size_t)n_pick * 8); memcpy(hp + n_pick, out_off, (size_t)(n_pick + 1) * 8);
    int64_t *d_pick = ctx->dp_pick.as<int64_t>(), *d_off = d_pick + n_pick;
    uint32_t *d_out = reinterpret_cast<uint32_t *>(d_off + n_pick + 1);
    HIPCHK(ctx, hipMemcpyAsync(d_pick, hp, (size_t)(2 * n_pick + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(dp_pick_cols, dim3((uint32_t)std::min<int64_t>((n_pick + 3) / 4, 256 * 8)), dim3(256), 0, ctx->stream, ctx->dp_cols.as<uint32_t>(), ctx->dp_last_col_off, d_pick,
                       d_off, n_pick, d_out);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out, d_out, (size_t)total * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MAUVE_OK;
}

int dp_batch_run(mauve_ctx *ctx, int nseq, int64_t n_iv, const uint8_t *codes, const int64_t *seq_off,
                 const mauve_scoring *scoring, uint32_t *cols, int64_t *col_off, int64_t *score, int64_t *cells)
{
    return dp_core(ctx, nseq, n_iv, codes, nullptr, seq_off, scoring, cols, col_off, score, cells);
}

static bool dp_desc_host() { static const bool h = getenv("MAUVE_DP_DESC_HOST") != nullptr; return h; }
static int64_t dp_desc_min() { static const int64_t m = getenv("MAUVE_DP_DESC_MIN") ? atoll(getenv("MAUVE_DP_DESC_MIN")) : 2048; return m; }
// does a batch of n_total rows take the device front (where the rotated candidates can be made on the device)?
bool dp_desc_rotations_on_device(int64_t n_total) { return !dp_desc_host() && n_total >= dp_desc_min(); }
int dp_batch_run_desc_rot(mauve_ctx *ctx, int nseq, int64_t n_orig, const DpSeqDesc *desc, const int32_t *cbase, int64_t n_total, const mauve_scoring *scoring,
                          int64_t *col_off, int64_t *score, int64_t *cells, int64_t *sp)
{
    if (!dp_desc_rotations_on_device(n_total)) { ctx->err = "dp: rotations on the device need the device front"; return MAUVE_ERR_STATE; }
    return dp_run_from_desc(ctx, nseq, n_total, desc, scoring, nullptr, col_off, score, cells, sp, n_orig, cbase);
}
int dp_batch_run_desc(mauve_ctx *ctx, int nseq, int64_t n_iv, const DpSeqDesc *desc, const mauve_scoring *scoring,
                      uint32_t *cols, int64_t *col_off, int64_t *score, int64_t *cells, bool may_shard, int64_t *sp)
{
    std::vector<int64_t> &seq_off = ctx->dph.seq_off;
    if (may_shard && ctx->shard_on && n_iv >= 2 * ctx->shard_world) {
        // Several contexts, one alignment (mauve_set_shard): the intervals are LPT-dealt by their cell bound, this rank aligns its
        // share, and everybody's columns, lengths and scores are exchanged -- [n, cells, len[n], score[n], cols...] per rank.
        std::vector<int64_t> cost((size_t)n_iv);
        for (int64_t k = 0; k < n_iv; k++) {
            int64_t m = 0, cl = 0;
            for (int g = 0; g < nseq; g++) { const int64_t n = desc[k * nseq + g].len; if (!n) continue; if (!m) { m = n; continue; } cl += m * n; m += n; }
            cost[(size_t)k] = cl;
        }
        std::vector<int> owner; shard_lpt(cost, ctx->shard_world, owner);
        std::vector<int64_t> mine;
        for (int64_t k = 0; k < n_iv; k++) if (owner[(size_t)k] == ctx->shard_rank) mine.push_back(k);
        const int64_t nmine = (int64_t)mine.size();
        std::vector<DpSeqDesc> sub((size_t)nmine * nseq);
        int64_t cap = 0;
        for (int64_t q = 0; q < nmine; q++)
            for (int g = 0; g < nseq; g++) { sub[(size_t)(q * nseq + g)] = desc[mine[(size_t)q] * nseq + g]; cap += desc[mine[(size_t)q] * nseq + g].len; }
        std::vector<uint32_t> mcols((size_t)cap + 1); std::vector<int64_t> moff((size_t)nmine + 1, 0), mscore((size_t)nmine + 1, 0), msp((size_t)nmine + 1, 0);
        int64_t mcells = 0;
        seq_off.resize((size_t)(nmine * nseq + 1));
        { int64_t t = 0; for (int64_t i = 0; i < nmine * nseq; i++) { seq_off[(size_t)i] = t; t += sub[(size_t)i].len; } seq_off[(size_t)(nmine * nseq)] = t; }
        const int rc_local = dp_core(ctx, nseq, nmine, nullptr, sub.data(), seq_off.data(), scoring, mcols.data(), moff.data(), mscore.data(), &mcells, sp ? msp.data() : nullptr);
        // a rank that failed still takes part in the exchange -- with the marker n = -1 -- so that the others do not wait in the collective for ever
        const int64_t ncol = rc_local ? 0 : moff[(size_t)nmine];
        std::vector<char> msg(rc_local ? 16 : (size_t)(2 + 3 * nmine) * 8 + (size_t)ncol * 4);         // [n, cells, len[n], score[n], sp[n], cols...]
        int64_t *h = reinterpret_cast<int64_t *>(msg.data());
        if (rc_local) { h[0] = -1; h[1] = rc_local; }
        else {
            h[0] = nmine; h[1] = mcells;
            for (int64_t q = 0; q < nmine; q++) { h[2 + q] = moff[(size_t)q + 1] - moff[(size_t)q]; h[2 + nmine + q] = mscore[(size_t)q]; h[2 + 2 * nmine + q] = msp[(size_t)q]; }
            if (ncol) memcpy(msg.data() + (size_t)(2 + 3 * nmine) * 8, mcols.data(), (size_t)ncol * 4);
        }
        std::vector<std::pair<const char *, size_t>> parts;
        const std::string err_local = ctx->err;
        int rc = shard_allgather(ctx, msg.data(), msg.size(), parts);
        if (rc) return rc;
        if (rc_local) { ctx->err = err_local; return rc_local; }
        for (int r = 0; r < ctx->shard_world; r++)
            if (parts[(size_t)r].second >= 16 && reinterpret_cast<const int64_t *>(parts[(size_t)r].first)[0] == -1) {
                ctx->err = "dp shard: rank " + std::to_string(r) + " failed (status " + std::to_string(reinterpret_cast<const int64_t *>(parts[(size_t)r].first)[1]) + ")";
                return MAUVE_ERR_STATE;
            }
        // lengths first (the offsets of ALL intervals in table order), then every rank's columns to their places
        std::vector<int64_t> len_of((size_t)n_iv, 0);
        std::vector<std::vector<int64_t>> ids((size_t)ctx->shard_world);
        for (int64_t k = 0; k < n_iv; k++) ids[(size_t)owner[(size_t)k]].push_back(k);
        int64_t total_cells = 0;
        for (int r = 0; r < ctx->shard_world; r++) {
            const int64_t *hr = reinterpret_cast<const int64_t *>(parts[(size_t)r].first);
            if (parts[(size_t)r].second < 16 || hr[0] != (int64_t)ids[(size_t)r].size()) { ctx->err = "dp shard: ranks disagree about the interval table"; return MAUVE_ERR_STATE; }
            total_cells += hr[1];
            {   // the announced lengths must account for the whole part: a short one would be copied past its end below
                const size_t nr = ids[(size_t)r].size();
                if (parts[(size_t)r].second < (2 + 3 * nr) * 8) { ctx->err = "dp shard: a rank's part is shorter than its header"; return MAUVE_ERR_STATE; }
                int64_t sum = 0; bool neg = false;
                for (size_t q = 0; q < nr; q++) { neg |= hr[2 + q] < 0; sum += hr[2 + q]; }
                if (neg || parts[(size_t)r].second != (2 + 3 * nr) * 8 + (size_t)sum * 4) { ctx->err = "dp shard: a rank's part does not hold the columns it announces"; return MAUVE_ERR_STATE; }
            }
            for (size_t q = 0; q < ids[(size_t)r].size(); q++) { len_of[(size_t)ids[(size_t)r][q]] = hr[2 + q]; if (score) score[ids[(size_t)r][q]] = hr[2 + ids[(size_t)r].size() + q]; if (sp) sp[ids[(size_t)r][q]] = hr[2 + 2 * ids[(size_t)r].size() + q]; }
        }
        col_off[0] = 0;
        for (int64_t k = 0; k < n_iv; k++) col_off[k + 1] = col_off[k] + len_of[(size_t)k];
        for (int r = 0; r < ctx->shard_world; r++) {
            const int64_t nr = (int64_t)ids[(size_t)r].size();
            const uint32_t *cr = reinterpret_cast<const uint32_t *>(parts[(size_t)r].first + (size_t)(2 + 3 * nr) * 8);
            for (int64_t q = 0; q < nr; q++) {
                const int64_t k = ids[(size_t)r][(size_t)q];
                memcpy(cols + col_off[k], cr, (size_t)len_of[(size_t)k] * 4);
                cr += len_of[(size_t)k];
            }
        }
        if (cells) *cells = total_cells;
        return MAUVE_OK;
    }
    // large batches: sizing, order and result offsets on the device (dp_run_from_desc); small ones keep the host loops (fewer launches).
    // MAUVE_DP_DESC_HOST: A/B switch
    if (!dp_desc_host() && n_iv >= dp_desc_min()) return dp_run_from_desc(ctx, nseq, n_iv, desc, scoring, cols, col_off, score, cells, sp);
    seq_off.resize((size_t)(n_iv * nseq + 1));
    int64_t t = 0;
    for (int64_t i = 0; i < n_iv * nseq; i++) { seq_off[(size_t)i] = t; t += desc[i].len; }
    seq_off[(size_t)(n_iv * nseq)] = t;
    return dp_core(ctx, nseq, n_iv, nullptr, desc, seq_off.data(), scoring, cols, col_off, score, cells, sp);
}

extern "C" int mauve_dp_batch(mauve_ctx *ctx, int nseq, int64_t n_iv, const uint8_t *codes, const int64_t *seq_off,
                              const mauve_scoring *sc, uint32_t *cols, int64_t *col_off, int64_t *score)
{
    if (!ctx) return MAUVE_ERR_ARG;
    if (nseq < 1 || nseq > MAUVE_MAX_SEQ || n_iv < 0 || !seq_off || !sc || !col_off || (n_iv && !cols)) {
        ctx->err = "dp_batch: bad argument"; return MAUVE_ERR_ARG;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->dp_band_from = INT64_MAX;          // the GappedAligner seam aligns what it is given in full
    return dp_batch_run(ctx, nseq, n_iv, codes, seq_off, sc, cols, col_off, score, nullptr);
}

extern "C" int mauve_dp_batch_banded(mauve_ctx *ctx, int nseq, int64_t n_iv, const uint8_t *codes, const int64_t *seq_off,
                                     const mauve_scoring *sc, int64_t band_from, uint32_t *cols, int64_t *col_off, int64_t *score)
{
    if (!ctx) return MAUVE_ERR_ARG;
    if (nseq < 1 || nseq > MAUVE_MAX_SEQ || n_iv < 0 || !seq_off || !sc || !col_off || (n_iv && !cols) || band_from < 0) {
        ctx->err = "dp_batch_banded: bad argument"; return MAUVE_ERR_ARG;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->dp_band_from = band_from;
    return dp_batch_run(ctx, nseq, n_iv, codes, seq_off, sc, cols, col_off, score, nullptr);
}
