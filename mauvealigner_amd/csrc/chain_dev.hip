// chain_dev.hip -- the chaining stage on the device: EliminateOverlaps and the LCB graph of Aligner::align [EXT]
// (call sites mauveAligner.cpp:596,600,698; helper usage toGrimmFormat.cpp:51-79, projectAndStrip.cpp:110-112),
// frozen spec DESIGN.md S5, for the N-way match list the seed pass leaves in HBM in canonical order.
//
// Shape: the list is a few 10^4 .. 10^5 records and the elimination is a chain of dependent passes (genome after
// genome, pass after pass).  The passes of one genome decompose into independent overlap clusters (see
// ch_cluster_pass), so a genome costs a fixed, short sequence of launches -- its order by the radix sort of
// seed_pass.hip, a prefix maximum of the right ends, the cluster starts, one thread per cluster for ALL its passes,
// the survivors' order -- with no launch per pass and no host round trip.  The LCB graph (collinear runs in genome-0
// order, their weights, the per-genome neighbour lists) is built by flag compactions; the greedy breakpoint
// elimination over that compact graph (10^2 .. 10^4 nodes, inherently sequential) stays on the host (lcb_greedy,
// chain_host.cpp).  Every kernel is tiled (1024 entries per workgroup, four consecutive entries per thread); the
// per-tile aggregates of a compaction are summed by each workgroup for itself, which saves the scan launches.
// The host code of chain_host.cpp remains the path for lists the seed pass sorted on the host (small ones) and for the small
// per-gap chains of the recursion.  A list whose canonical order has ties (equal first component and start) does come here:
// the seed pass finishes the order inside the tie groups on the host and writes the repaired list back to sorted_rec before it
// sets dev_rec_n, and every pass below breaks equal left ends by the list index, exactly as chain_host.cpp does.
#include "common.hpp"
#include "dev_scan.hpp"
#include <cstring>
#include <cstdlib>

namespace {

using namespace devscan;
constexpr int CH_TILE = devscan::TILE;

// records as the seed pass left them (int64 length[n], start[n*N]) -> working arrays (int32)
// (both init kernels also clear what the stage accumulates into -- the counter block and the node weights -- instead of two memsets)
__global__ void __launch_bounds__(256) ch_init(const int64_t *__restrict__ rlen, const int64_t *__restrict__ rst, uint32_t n, int N,
                                               int32_t *__restrict__ len, int32_t *__restrict__ st, uint32_t *__restrict__ crop,
                                               uint32_t *__restrict__ cnt, unsigned long long *__restrict__ weight)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < 64) cnt[i] = 0;
    if (i >= n) return;
    weight[i] = 0;
    len[i] = (int32_t)rlen[i];
    for (int g = 0; g < N; g++) st[(size_t)i * N + g] = (int32_t)rst[(size_t)i * N + g];
    crop[2 * (size_t)i] = 0; crop[2 * (size_t)i + 1] = 0;
}

// the same for the N-way list of a recursion batch (virtual genomes = the gaps of the batch, concatenated): only
// all-forward matches take part, and every match gets the id of its gap -- the segment of genome 0 its
// start lies in (seg0: K + 1 segment starts)
__global__ void __launch_bounds__(256) ch_init_seg(const int64_t *__restrict__ rlen, const int64_t *__restrict__ rst, uint32_t n, int N,
                                                   const uint32_t *__restrict__ seg0, uint32_t K, int32_t *__restrict__ len, int32_t *__restrict__ st,
                                                   uint32_t *__restrict__ crop, uint32_t *__restrict__ gapid,
                                                   uint32_t *__restrict__ cnt, unsigned long long *__restrict__ weight)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < 64) cnt[i] = 0;
    if (i >= n) return;
    weight[i] = 0;
    bool fwd = true;
    for (int g = 0; g < N; g++) { const int64_t s = rst[(size_t)i * N + g]; st[(size_t)i * N + g] = (int32_t)s; fwd &= s > 0; }
    len[i] = fwd ? (int32_t)rlen[i] : 0;
    crop[2 * (size_t)i] = 0; crop[2 * (size_t)i + 1] = 0;
    const uint32_t p0 = (uint32_t)(rst[(size_t)i * N] > 0 ? rst[(size_t)i * N] - 1 : 0);
    uint32_t lo = 0, hi = K;                                     // last k with seg0[k] <= p0
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (seg0[mid] <= p0) lo = mid; else hi = mid; }
    gapid[i] = lo;
}

// sort keys of genome g: left end of every alive match (dead ones sort behind everything), value = match index
__global__ void __launch_bounds__(256) ch_keys(const int32_t *__restrict__ len, const int32_t *__restrict__ st, uint32_t n, int N, int g,
                                               uint32_t dead_key, uint32_t *__restrict__ key, uint32_t *__restrict__ val)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const int32_t s = st[(size_t)i * N + g];
    key[i] = len[i] > 0 ? (uint32_t)(s < 0 ? -s : s) : dead_key;
    val[i] = i;
}

// ---- EliminateOverlaps of one genome, decomposed into overlap clusters ----------------------------
// The passes of genome g only ever shrink intervals in g, so two matches interact only if their intervals are linked
// through a chain of overlaps when g's turn starts: cut the (left end, index) order wherever the running maximum of
// the right ends stays below the next left end, and the pieces -- clusters -- never see each other again: the global
// order is the concatenation of the clusters' orders and no pair across a cut ever overlaps.  Every cluster
// therefore runs ALL its passes on its own (sort, sweep of adjacent pairs, crops applied together, the dead leave,
// repeat until overlap free), exactly as the whole list would, and the clusters run side by side, one thread each:
// no pass-by-pass launches, no host round trip.  Nearly all clusters are pairs; one beyond CH_CL_MAX entries (a
// repeat family) raises *fail and the host chains this list with chain_host.cpp instead.
constexpr int CH_CL_MAX = 48;

// right end of entry r (0 for the dead entries behind the alive ones and beyond the list)
__device__ __forceinline__ uint32_t ch_right(const int32_t *__restrict__ len, uint32_t n, uint32_t dead_key,
                                             const uint32_t *__restrict__ eleft, const uint32_t *__restrict__ eidx, uint32_t r)
{
    if (r >= n) return 0u;
    const uint32_t l = eleft[r];
    return l == dead_key ? 0u : l + (uint32_t)len[eidx[r]] - 1u;
}

// per tile: the largest right end and the number of alive entries
__global__ void __launch_bounds__(256) cl_partial(const int32_t *__restrict__ len, uint32_t n, uint32_t dead_key, const uint32_t *__restrict__ eleft,
                                                  const uint32_t *__restrict__ eidx, uint32_t *__restrict__ bmax, uint32_t *__restrict__ balive)
{
    __shared__ uint32_t lds[4];
    const uint32_t r0 = blockIdx.x * (uint32_t)CH_TILE + threadIdx.x * 4u;
    uint32_t mx = 0, c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t rt = ch_right(len, n, dead_key, eleft, eidx, r0 + i);
        mx = max(mx, rt); c += rt != 0;
    }
    uint32_t tm, tc;
    (void)bscan_max(mx, &tm, lds); (void)bscan_add(c, &tc, lds);
    if (threadIdx.x == 0) { bmax[blockIdx.x] = tm; balive[blockIdx.x] = tc; }
}

// cluster starts: entry r starts a cluster iff every right end before it lies left of its left end
__global__ void __launch_bounds__(256) cl_flags(const int32_t *__restrict__ len, uint32_t n, uint32_t dead_key, const uint32_t *__restrict__ eleft,
                                                const uint32_t *__restrict__ eidx, const uint32_t *__restrict__ bmax,
                                                const uint32_t *__restrict__ balive, uint32_t nb, uint8_t *__restrict__ cflag,
                                                uint32_t *__restrict__ cnt, uint32_t *__restrict__ bcnt, uint32_t *__restrict__ alive_out)
{
    __shared__ uint32_t lds[4];
    const uint32_t b = blockIdx.x, r0 = b * (uint32_t)CH_TILE + threadIdx.x * 4u;
    uint32_t mb = 0;
    for (uint32_t t = threadIdx.x; t < b; t += 256) mb = max(mb, bmax[t]);
    uint32_t before;
    (void)bscan_max(mb, &before, lds);
    uint32_t rt[4], mx = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { rt[i] = ch_right(len, n, dead_key, eleft, eidx, r0 + i); mx = max(mx, rt[i]); }
    uint32_t dummy;
    uint32_t run = max(before, bscan_max(mx, &dummy, lds));
    uint32_t nflag = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t r = r0 + i;
        if (r < n) { const bool f = rt[i] != 0 && (r == 0 || run < eleft[r]); cflag[r] = f ? 1 : 0; nflag += f ? 1u : 0u; }
        run = max(run, rt[i]);
    }
    {   // the tile's flag count, for the compaction of the cluster starts that follows (what cmp_count<ClusterStarts> would launch for)
        uint32_t tc;
        (void)bscan_add(nflag, &tc, lds);
        if (threadIdx.x == 0) bcnt[b] = tc;
    }
    if (b == 0) {                                        // alive entries of the genome in turn -> cnt[5]
        uint32_t sa = 0, k;
        for (uint32_t t = threadIdx.x; t < nb; t += 256) sa += balive[t];
        (void)bscan_add(sa, &k, lds);
        if (threadIdx.x == 0) { cnt[5] = k; *alive_out = k; }      // (alive_out: kept per genome, the domain of FinalRanks)
    }
}

struct ClusterStarts {                      // flagged entries -> cstart[]; cstart[J] = k
    const uint8_t *cflag; uint32_t n; uint32_t *cstart; uint32_t *cnt;
    __device__ uint32_t domain(int) const { return n; }
    __device__ bool flag(uint32_t r, int) const { return cflag[r] != 0; }
    __device__ void each(uint32_t, uint32_t, bool, int) const {}
    __device__ void emit(uint32_t r, uint32_t o, int) const { cstart[o] = r; }
    __device__ void total(uint32_t J, int) const { cnt[4] = J; cstart[J] = cnt[5]; }
};
struct FinalRanks {                         // y = genome: its order (as its cluster pass left it: the entries alive BEFORE that pass, those it killed marked by
                                            // the dead key) without the matches that died then or later; rank of every survivor
    const int32_t *len; uint32_t n; const uint32_t *ord, *okey; uint32_t dead_key; uint32_t *ordc, *rank; uint32_t *cnt;
    __device__ uint32_t domain(int y) const { return cnt[8 + y]; }
    __device__ bool flag(uint32_t r, int y) const { return okey[(size_t)y * n + r] != dead_key && len[ord[(size_t)y * n + r]] > 0; }
    __device__ void each(uint32_t, uint32_t, bool, int) const {}
    __device__ void emit(uint32_t r, uint32_t o, int y) const { const uint32_t i = ord[(size_t)y * n + r]; ordc[(size_t)y * n + o] = i; rank[(size_t)y * n + i] = o; }
    __device__ void total(uint32_t na, int y) const { if (y == 0) cnt[0] = na; }
};
// LCB nodes = maximal collinear runs in genome-0 order: two neighbours belong together iff in every other genome
// they have the same orientation and are neighbours there too (next if forward, previous if reverse)
struct LcbNodes {
    const int32_t *len, *st; uint32_t n; int N; const uint32_t *ordc, *rank; uint32_t *cnt; int32_t *node_of;
    unsigned long long *weight; uint32_t *orient; const uint32_t *gapid;        // gapid (recursion batches): a node never spans two gaps
    const int64_t *mw;                                                            // per-match weights (sum-of-pairs scoring), or null
    __device__ uint32_t domain(int) const { return cnt[0]; }
    __device__ bool flag(uint32_t k, int) const
    {
        if (k == 0) return true;
        const uint32_t i = ordc[k], p = ordc[k - 1];
        if (gapid && gapid[i] != gapid[p]) return true;
        // (no early exit: the genomes' loads are independent and travel together; with a return per genome they were a chain of N - 1 round trips)
        bool brk = false;
#pragma unroll 4
        for (int g = 1; g < N; g++) {
            const bool oi = st[(size_t)i * N + g] < 0, op = st[(size_t)p * N + g] < 0;
            const uint32_t ri = rank[(size_t)g * n + i], rp = rank[(size_t)g * n + p];
            brk |= (oi != op) | (!oi ? ri != rp + 1 : ri + 1 != rp);
        }
        return brk;
    }
    __device__ void each(uint32_t k, uint32_t o, bool fl, int) const
    {
        const uint32_t i = ordc[k], nd = o + (fl ? 1u : 0u) - 1u;
        node_of[i] = (int32_t)nd;
        // LCB weight: columns x genomes (Aligner::align), or the matches' extant sum-of-pairs scores (two's complement: a sum of signed scores)
        atomicAdd(&weight[nd], mw ? (unsigned long long)mw[i] : (unsigned long long)len[i] * (unsigned long long)N);
    }
    __device__ void emit(uint32_t k, uint32_t o, int) const
    {
        const uint32_t i = ordc[k];
        uint32_t ob = 0;
        for (int g = 0; g < N; g++) if (st[(size_t)i * N + g] < 0) ob |= 1u << g;
        orient[o] = ob;
    }
    __device__ void total(uint32_t K, int) const { cnt[1] = K; }
};
struct NodeSeq {                            // y = genome: the nodes in genome-y order (a node's matches are contiguous in every genome)
    uint32_t n; const uint32_t *ordc; uint32_t *cnt; const int32_t *node_of; int32_t *seq;
    __device__ uint32_t domain(int) const { return cnt[0]; }
    __device__ bool flag(uint32_t r, int y) const { return r == 0 || node_of[ordc[(size_t)y * n + r]] != node_of[ordc[(size_t)y * n + r - 1]]; }
    __device__ void each(uint32_t, uint32_t, bool, int) const {}
    __device__ void emit(uint32_t r, uint32_t o, int y) const { if (o < cnt[1]) seq[(size_t)y * cnt[1] + o] = node_of[ordc[(size_t)y * n + r]]; }
    __device__ void total(uint32_t t, int) const { if (t != cnt[1]) atomicAdd(&cnt[2], 1u); }    // cannot happen (contiguity); the host checks
};
__global__ void __launch_bounds__(256) ch_links(int N, const uint32_t *__restrict__ cnt, const int32_t *__restrict__ seq,
                                                int32_t *__restrict__ prevv, int32_t *__restrict__ nextv)
{
    const uint32_t K = cnt[1], j = blockIdx.x * 256u + threadIdx.x;
    const int g = blockIdx.y;
    if (j >= K) return;
    const int32_t *sq = seq + (size_t)g * K;
    const int32_t nd = sq[j];
    prevv[(size_t)nd * N + g] = j > 0 ? sq[j - 1] : -1;
    nextv[(size_t)nd * N + g] = j + 1 < K ? sq[j + 1] : -1;
}

// ch_cluster_pass: thread j runs every pass of cluster j.  Entries come back in (left end, index) order, the dead
// behind them as dead_key; the survivors' crops are applied to the match records (all genomes) at the end.
// all passes of one cluster (entries a .. a + s of the genome's order); the working arrays hold s entries each
__device__ __forceinline__ void ch_cluster_run(int32_t *__restrict__ len, int32_t *__restrict__ st, int N, int g, uint32_t dead_key,
                                               uint32_t *__restrict__ eleft, uint32_t *__restrict__ eidx, uint32_t a, int s,
                                               uint32_t *L, uint32_t *Ln, uint32_t *I, uint32_t *CF, uint32_t *CL, uint32_t *pf, uint32_t *pl, uint8_t *F)
{
    // entry state: left end in g, length, index, forward in g, crops so far at the match's first / last column side
    for (int q = 0; q < s; q++) {
        L[q] = eleft[a + q]; I[q] = eidx[a + q]; Ln[q] = (uint32_t)len[I[q]]; F[q] = st[(size_t)I[q] * N + g] > 0;
        CF[q] = 0; CL[q] = 0;
    }
    const int s0 = s;
    for (;;) {
        // (left end, index) order; the entries arrive sorted and a pass of crops moves them little: insertion sort
        for (int q = 1; q < s; q++) {
            const uint32_t l = L[q], n2 = Ln[q], i2 = I[q], cf = CF[q], cl = CL[q]; const uint8_t f = F[q];
            int t = q;
            while (t > 0 && (L[t - 1] > l || (L[t - 1] == l && I[t - 1] > i2))) {
                L[t] = L[t - 1]; Ln[t] = Ln[t - 1]; I[t] = I[t - 1]; CF[t] = CF[t - 1]; CL[t] = CL[t - 1]; F[t] = F[t - 1]; t--;
            }
            L[t] = l; Ln[t] = n2; I[t] = i2; CF[t] = cf; CL[t] = cl; F[t] = f;
        }
        // sweep: of two overlapping neighbours the shorter (the right one on ties) gives up the overlap on the side
        // facing the other; lengths as of the pass start; every side gets at most one request
        bool any = false;
        for (int q = 0; q < s; q++) { pf[q] = 0; pl[q] = 0; }
        for (int q = 0; q + 1 < s; q++) {
            const int64_t ov = (int64_t)L[q] + Ln[q] - (int64_t)L[q + 1];
            if (ov <= 0) continue;
            any = true;
            if (Ln[q] < Ln[q + 1]) { if (F[q]) pl[q] = (uint32_t)ov; else pf[q] = (uint32_t)ov; }              // its right side in g
            else { if (F[q + 1]) pf[q + 1] = (uint32_t)ov; else pl[q + 1] = (uint32_t)ov; }                    // its left side in g
        }
        if (!any) break;
        // the crops of the pass together; a match whose requests reach its length dies
        int w = 0;
        for (int q = 0; q < s; q++) {
            const int64_t nl = (int64_t)Ln[q] - pf[q] - pl[q];
            if (nl <= 0) { len[I[q]] = 0; continue; }
            L[w] = L[q] + (F[q] ? pf[q] : pl[q]); Ln[w] = (uint32_t)nl; I[w] = I[q]; F[w] = F[q];
            CF[w] = CF[q] + pf[q]; CL[w] = CL[q] + pl[q];
            w++;
        }
        s = w;
    }
    for (int q = 0; q < s; q++) {
        eleft[a + q] = L[q]; eidx[a + q] = I[q];
        if (CF[q] | CL[q]) {
            const uint32_t i = I[q];
            for (int c = 0; c < N; c++) {
                const int32_t v = st[(size_t)i * N + c];
                st[(size_t)i * N + c] = v > 0 ? v + (int32_t)CF[q] : v - (int32_t)CL[q];
            }
            len[i] = (int32_t)Ln[q];
        }
    }
    for (int q = s; q < s0; q++) eleft[a + q] = dead_key;
}

// big == nullptr: a cluster beyond CH_CL_MAX entries raises *fail (the caller chains on the host instead).
__global__ void __launch_bounds__(256) ch_cluster_pass(int32_t *__restrict__ len, int32_t *__restrict__ st, int N, int g, uint32_t dead_key,
                                                       uint32_t *__restrict__ eleft, uint32_t *__restrict__ eidx,
                                                       const uint32_t *__restrict__ cstart, const uint32_t *__restrict__ cnt_in,
                                                       uint32_t *__restrict__ fail, int leave_big, int cl_max)
{
    // Nearly every thread has a cluster of one and leaves at once; the few that work keep their arrays in LDS (a slot each,
    // handed out by a counter; sized by the cluster: 8 words per entry) instead of in scratch, whose latency was the kernel's
    // whole run time.  A block that runs out of LDS falls back to scratch for the rest.
    constexpr int POOL_WORDS = 12 * 1024;                      // 48 KB
    __shared__ uint32_t pool[POOL_WORDS];
    __shared__ uint32_t pool_used;
    if (threadIdx.x == 0) pool_used = 0;
    __syncthreads();
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= cnt_in[0]) return;
    const uint32_t a = cstart[j];
    const int s = (int)(cstart[j + 1] - a);
    if (s < 2) return;
    if (s > cl_max) { if (!leave_big) *fail = 1; return; }
    const uint32_t want = 8u * (uint32_t)s, at = atomicAdd(&pool_used, want);
    if (at + want <= (uint32_t)POOL_WORDS) {
        uint32_t *w = pool + at;
        ch_cluster_run(len, st, N, g, dead_key, eleft, eidx, a, s, w, w + s, w + 2 * s, w + 3 * s, w + 4 * s, w + 5 * s, w + 6 * s,
                       reinterpret_cast<uint8_t *>(w + 7 * s));
        return;
    }
    uint32_t L[CH_CL_MAX], Ln[CH_CL_MAX], I[CH_CL_MAX], CF[CH_CL_MAX], CL[CH_CL_MAX], pf[CH_CL_MAX], pl[CH_CL_MAX];
    uint8_t F[CH_CL_MAX];
    ch_cluster_run(len, st, N, g, dead_key, eleft, eidx, a, s, L, Ln, I, CF, CL, pf, pl, F);
}

// The clusters ch_cluster_pass left alone (recursion batches: light seeds in divergent stretches give repeat-like families of
// overlapping matches): the same passes with the working arrays in global memory (ws: 7 words + 1 byte per list entry,
// a cluster uses the slice of its own entries), still one thread per cluster -- they are few.
__global__ void __launch_bounds__(64) ch_cluster_pass_big(int32_t *__restrict__ len, int32_t *__restrict__ st, int N, int g, uint32_t dead_key,
                                                          uint32_t *__restrict__ eleft, uint32_t *__restrict__ eidx,
                                                          const uint32_t *__restrict__ cstart, const uint32_t *__restrict__ cnt_in,
                                                          uint32_t *__restrict__ ws, uint32_t n, int cl_max, uint32_t *__restrict__ fail, int big_max)
{
    const uint32_t j = blockIdx.x * 64u + threadIdx.x;
    if (j >= cnt_in[0]) return;
    const uint32_t a = cstart[j];
    const int s = (int)(cstart[j + 1] - a);
    if (s <= cl_max) return;
    // one lane walks the whole cluster, every pass an insertion sort over global memory: beyond a few thousand entries that is seconds of
    // one dependent chain (it would look like a hang on the stream) -- such a list goes back to the host's per-gap chaining instead
    if (s > big_max) { *fail = 1; return; }
    uint32_t *L = ws + a, *Ln = L + n, *I = Ln + n, *CF = I + n, *CL = CF + n, *pf = CL + n, *pl = pf + n;
    uint8_t *F = reinterpret_cast<uint8_t *>(ws + 7 * (size_t)n) + a;
    ch_cluster_run(len, st, N, g, dead_key, eleft, eidx, a, s, L, Ln, I, CF, CL, pf, pl, F);
}

// the compact graph for the host's greedy elimination, in one piece: weight[K] (8 B), orient[K], prev[K*N], next[K*N], check word
__global__ void __launch_bounds__(256) ch_pack_graph(uint32_t K, int N, const unsigned long long *__restrict__ weight, const uint32_t *__restrict__ orient,
                                                     const int32_t *__restrict__ prevv, const int32_t *__restrict__ nextv, const uint32_t *__restrict__ cnt,
                                                     unsigned long long *__restrict__ out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint32_t *o32 = reinterpret_cast<uint32_t *>(out + K);
    int32_t *op = reinterpret_cast<int32_t *>(o32 + K), *on = op + (size_t)K * N;
    if (i < K) { out[i] = weight[i]; o32[i] = orient[i]; }
    if (i < K * (uint32_t)N) { op[i] = prevv[i]; on[i] = nextv[i]; }
    if (i == 0) reinterpret_cast<uint32_t *>(on + (size_t)K * N)[0] = cnt[2];
}

// final LCB id of every match (-1: dead, or its LCB was eliminated)
__global__ void __launch_bounds__(256) ch_label(const int32_t *__restrict__ len, uint32_t n, const int32_t *__restrict__ node_of,
                                                const int32_t *__restrict__ final_id, int32_t *__restrict__ lcb)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    lcb[i] = len[i] > 0 ? final_id[node_of[i]] : -1;
}

// ---- the chains on the device (align_device_tail, pipeline.cpp) ---------------------------------------------------
// chain order = LCB by LCB, genome-0 order inside: a stable sort by LCB id of the survivors taken in their genome-0 order AFTER the
// elimination (ord0: the crops of a pass can carry a match past a neighbour, DESIGN.md S5 -- the list index is not that order)
__global__ void __launch_bounds__(256) co_keys(const int32_t *__restrict__ lcb, const uint32_t *__restrict__ ord0, const uint32_t *__restrict__ cnt, uint32_t n, uint32_t nl,
                                               uint32_t *__restrict__ key, uint32_t *__restrict__ val, unsigned long long *__restrict__ lw, uint32_t *__restrict__ out)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k < nl) lw[k] = 0;                                    // what co_gather accumulates into (nl <= n)
    if (k < 2) out[k] = 0;
    if (k >= n) return;
    if (k < cnt[0]) {
        const uint32_t i = ord0[k];
        const int32_t l = lcb[i];
        key[k] = l >= 0 ? (uint32_t)l : nl;                  // eliminated with its LCB: behind every LCB
        val[k] = i;
    } else { key[k] = nl; val[k] = 0; }                      // (dead matches are not in ord0)
}

// anchors in chain order (alen[cap], ast[cap * N], alcb[cap]), their number, LCB weights (sum of length * N), and the
// number of inter-anchor gaps the recursion would have to look at (longest side above min_gap; gap_weight, recursive.cpp)
__global__ void __launch_bounds__(256) co_gather(const uint32_t *__restrict__ key, const uint32_t *__restrict__ val, uint32_t n, uint32_t nl, int N,
                                                 const int32_t *__restrict__ len, const int32_t *__restrict__ st, int64_t min_gap,
                                                 int32_t *__restrict__ alen, int32_t *__restrict__ ast, int32_t *__restrict__ alcb,
                                                 unsigned long long *__restrict__ lw, uint32_t *__restrict__ out)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    const uint32_t l = j < n ? key[j] : nl;
    const bool alive = l < nl;
    const uint32_t i = alive ? val[j] : 0u;
    const int32_t ln = alive ? len[i] : 0;
    wave_keyed_add(lw, alive, l, (unsigned long long)((int64_t)ln * N));
    if (!alive) return;
    alen[j] = ln; alcb[j] = (int32_t)l;
    for (int g = 0; g < N; g++) ast[(size_t)j * N + g] = st[(size_t)i * N + g];
    const bool last = j + 1 == n || key[j + 1] >= nl;
    if (last) out[0] = j + 1;                                // the labelled matches are the first na entries of the order
    if (!last && key[j + 1] == l) {
        const uint32_t i2 = val[j + 1];
        const int32_t ln2 = len[i2];
        int64_t mx = 0;
        for (int g = 0; g < N; g++) {
            const int64_t sa = st[(size_t)i * N + g], sb = st[(size_t)i2 * N + g];
            int64_t lo, hi;
            if (sa > 0) { lo = sa + ln; hi = sb - 1; } else { lo = -sb + ln2; hi = -sa - 1; }
            mx = max(mx, hi - lo + 1);
        }
        if (mx > min_gap) atomicAdd(&out[1], 1u);
    }
}


// ---- the collinear rule of a recursion batch on the device (one collinear chain per gap) ----
// Every gap's nodes are consecutive node ids (genome-0 order).  gap_node_gap: the gap of a node, from any of its matches;
// gap_ranges: the node range of every gap that holds more than one node (the others survive as they are); gap_greedy: one wave
// per such gap runs lcb_greedy's collinear form on the gap's sub-graph in LDS -- the lightest node (lowest id on ties) is
// deleted and its neighbours re-merged, until one node is left -- and marks the nodes that end up in the survivor.
__global__ void __launch_bounds__(256) gap_node_gap(const int32_t *__restrict__ len, const int32_t *__restrict__ node_of, const uint32_t *__restrict__ gapid, uint32_t n,
                                                    uint32_t *__restrict__ node_gap)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n && len[i] > 0) node_gap[node_of[i]] = gapid[i];
}
__global__ void __launch_bounds__(256) gap_ranges(const uint32_t *__restrict__ node_gap, uint32_t K, uint8_t *__restrict__ node_ok, uint32_t *__restrict__ ranges /* pairs */,
                                                  uint32_t *__restrict__ count)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= K) return;
    node_ok[j] = 1;                                               // (gap_greedy clears what the rule drops)
    const uint32_t g = node_gap[j];
    if (j > 0 && node_gap[j - 1] == g) return;                    // not the first node of its gap
    uint32_t e = j + 1;
    while (e < K && node_gap[e] == g) e++;
    if (e - j > 1) { const uint32_t o = atomicAdd(count, 1u); ranges[2 * (size_t)o] = j; ranges[2 * (size_t)o + 1] = e; }
}
constexpr int GAP_KMAX = 512;                                     // nodes of one gap the wave keeps in LDS (more: the host does that batch)
constexpr int GAP_LINKS = 2560;                                   // ... and node x genome link entries
__global__ void __launch_bounds__(64) gap_greedy(const uint32_t *__restrict__ ranges, const uint32_t *__restrict__ nranges_p, int N, const unsigned long long *__restrict__ weight,
                                                 const int32_t *__restrict__ prevv, const int32_t *__restrict__ nextv, uint8_t *__restrict__ node_ok,
                                                 uint32_t *__restrict__ fail)
{
    __shared__ unsigned long long s_w[GAP_KMAX];
    __shared__ int16_t s_pv[GAP_LINKS], s_nx[GAP_LINKS], s_merged[GAP_KMAX];
    __shared__ uint8_t s_alive[GAP_KMAX];
    const int lane = threadIdx.x;
    const uint32_t nranges = *nranges_p;                          // (a device value: the launch is sized for the worst case)
    for (uint32_t r = blockIdx.x; r < nranges; r += gridDim.x) {
        const uint32_t n0 = ranges[2 * (size_t)r], n1 = ranges[2 * (size_t)r + 1];
        const int kk = (int)(n1 - n0);
        if (kk > GAP_KMAX || kk * N > GAP_LINKS) { if (lane == 0) atomicAdd(fail, 1u); continue; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        for (int j = lane; j < kk; j += 64) { s_w[j] = weight[n0 + j]; s_alive[j] = 1; s_merged[j] = -1; }
        for (int t = lane; t < kk * N; t += 64) {
            const int32_t p = prevv[(size_t)n0 * N + t], q = nextv[(size_t)n0 * N + t];
            s_pv[t] = (p >= (int32_t)n0 && p < (int32_t)n1) ? (int16_t)(p - (int32_t)n0) : (int16_t)-1;
            s_nx[t] = (q >= (int32_t)n0 && q < (int32_t)n1) ? (int16_t)(q - (int32_t)n0) : (int16_t)-1;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        int alive_cnt = kk;
        while (alive_cnt > 1) {
            // the lightest alive node, lowest id on ties (weights are sums of match lengths x N: far below 2^48)
            unsigned long long best = ~0ULL;
            for (int j = lane; j < kk; j += 64) if (s_alive[j]) { const unsigned long long key = (s_w[j] << 16) | (unsigned long long)j; best = key < best ? key : best; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(best, o); best = t < best ? t : best; }
            const int x = (int)(best & 0xffffULL);
            if (lane == 0) {
                // unlink x, then let its former neighbours merge where they have become collinear neighbours (all matches of a batch are forward)
                int16_t ca[MAUVE_MAX_SEQ], cb[MAUVE_MAX_SEQ];
                for (int g = 0; g < N; g++) { ca[g] = s_pv[x * N + g]; cb[g] = s_nx[x * N + g]; }
                for (int g = 0; g < N; g++) {
                    const int p = s_pv[x * N + g], q = s_nx[x * N + g];
                    if (p >= 0) s_nx[p * N + g] = (int16_t)q;
                    if (q >= 0) s_pv[q * N + g] = (int16_t)p;
                }
                s_alive[x] = 0; alive_cnt--;
                for (int g = 0; g < N; g++) {
                    int a = ca[g], b = cb[g];
                    while (a >= 0 && !s_alive[a] && s_merged[a] >= 0) a = s_merged[a];
                    while (b >= 0 && !s_alive[b] && s_merged[b] >= 0) b = s_merged[b];
                    if (a < 0 || b < 0 || a == b || !s_alive[a] || !s_alive[b]) continue;
                    if (s_nx[b * N] == a) { const int t = a; a = b; b = t; }
                    if (s_nx[a * N] != b) continue;
                    bool ok = true;
                    for (int h = 1; h < N && ok; h++) ok = s_nx[a * N + h] == b;
                    if (!ok) continue;
                    s_w[a] += s_w[b];
                    for (int h = 0; h < N; h++) {
                        const int p = s_pv[b * N + h], q = s_nx[b * N + h];
                        if (p >= 0) s_nx[p * N + h] = (int16_t)q;
                        if (q >= 0) s_pv[q * N + h] = (int16_t)p;
                    }
                    s_alive[b] = 0; s_merged[b] = (int16_t)a; alive_cnt--;
                }
            }
            alive_cnt = __shfl(alive_cnt, 0);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        }
        for (int j = lane; j < kk; j += 64) {
            int v = j;
            while (!s_alive[v] && s_merged[v] >= 0) v = s_merged[v];
            node_ok[n0 + j] = s_alive[v] ? 1 : 0;
        }
    }
}
// the survivors of a recursion batch, in list order: record (length, starts), gap id
struct GapSurvivors {
    const int32_t *len, *st, *node_of; const uint32_t *gapid; const uint8_t *node_ok; uint32_t n; int N;
    int32_t *olen, *ost; uint32_t *ogap; uint32_t *total_out;
    __device__ uint32_t domain(int) const { return n; }
    __device__ bool flag(uint32_t i, int) const { return len[i] > 0 && node_ok[node_of[i]]; }
    __device__ void each(uint32_t, uint32_t, bool, int) const {}
    __device__ void emit(uint32_t i, uint32_t o, int) const
    {
        olen[o] = len[i]; ogap[o] = gapid[i];
        for (int g = 0; g < N; g++) ost[(size_t)o * N + g] = st[(size_t)i * N + g];
    }
    __device__ void total(uint32_t t, int) const { *total_out = t; }
};

}  // namespace

// The chains stay on the device: anchors in chain order in c->ch_anch (capacity layout: alen[n], ast[n * N], alcb[n]),
// LCB weights in c->ch_lw.  Out: number of anchors, number of gaps the recursion would look at.
int chain_order_device(mauve_ctx *c, int N, int64_t nl, int64_t min_gap, int64_t *na_out, int64_t *n_rec_out)
{
    const uint32_t n = (uint32_t)c->dev_rec_n;
    *na_out = 0; *n_rec_out = 0;
    if (nl <= 0) return MAUVE_OK;
    const int32_t *len = c->ch_len.as<int32_t>(), *lcb = len + 2 * (size_t)n;
    const int32_t *st = c->ch_st.as<int32_t>();
    uint32_t *k1 = c->ch_ent.as<uint32_t>(), *v1 = k1 + n, *k2 = v1 + n, *v2 = k2 + n;
    HIPCHK(c, c->ch_anch.ensure((size_t)n * (2 + (size_t)N) * 4 + 64));
    HIPCHK(c, c->ch_lw.ensure((size_t)nl * 8 + 64));
    int32_t *alen = c->ch_anch.as<int32_t>(), *ast = alen + n, *alcb = ast + (size_t)n * N;
    unsigned long long *lw = c->ch_lw.as<unsigned long long>();
    uint32_t *cnt = c->ch_cnt.as<uint32_t>();
    const uint32_t blocks = (n + 255) / 256;
    int bits = 1; while ((1LL << bits) <= nl) bits++;
    hipLaunchKernelGGL(co_keys, dim3(blocks), dim3(256), 0, c->stream, lcb, c->ch_ord.as<uint32_t>() + (size_t)n * N /* ordc of genome 0 */, cnt, n, (uint32_t)nl, k1, v1, lw, cnt + 16);
    uint32_t *kk = k1, *vv = v1;
    int rc = sort_pairs_u32(c, n, bits, &kk, &vv, k2, v2, MAUVE_K_MISC);
    if (rc) return rc;
    hipLaunchKernelGGL(co_gather, dim3(blocks), dim3(256), 0, c->stream, kk, vv, n, (uint32_t)nl, N, len, st, min_gap, alen, ast, alcb, lw, cnt + 16);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, c->pin_chain.ensure(256));
    HIPCHK(c, hipMemcpyAsync(c->pin_chain.p, cnt + 16, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *na_out = c->pin_chain.as<uint32_t>()[0];
    *n_rec_out = c->pin_chain.as<uint32_t>()[1];
    return MAUVE_OK;
}

// Elimination and LCB graph of the device-resident list (cropped records in c->ch_len / c->ch_st, node of every match behind
// the lengths).  The compact graph arrives on the host in c->pin_chain (ChainGraphHost): weight[K], orient[K], prev[K*N],
// next[K*N].  seg0 != nullptr: a recursion batch (forward matches only, nodes confined to their gaps).
// extant sum-of-pairs score of every cropped record of the chain (sp_score_matches' rule on the chain's int32 records): a wave per match,
// a lane takes every 64th column; dead records (length <= 0) score 0
struct ChSpGenomes { uint64_t word_off[MAUVE_MAX_SEQ]; int32_t s[4][4]; };
__global__ void __launch_bounds__(256) ch_sp_scores(const uint64_t *__restrict__ packed, ChSpGenomes G, int N, const int32_t *__restrict__ len, const int32_t *__restrict__ st,
                                                    uint32_t n, int64_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nw = (gridDim.x * 256u) >> 6;
    for (uint32_t i = wave; i < n; i += nw) {
        const int64_t L = len[i];
        int64_t acc = 0;
        for (int64_t c = lane; c < L; c += 64) {
            uint32_t have = 0, bases = 0;
            for (int g = 0; g < N; g++) {
                const int64_t s0 = st[(size_t)i * N + g];
                if (!s0) continue;
                const int64_t p = s0 > 0 ? s0 - 1 + c : -s0 - 1 + (L - 1 - c);
                uint32_t b = (uint32_t)(packed[G.word_off[g] + (uint64_t)(p >> 5)] >> (2 * (p & 31))) & 3u;
                if (s0 < 0) b = 3u - b;
                have |= 1u << g; bases |= b << (2 * g);
            }
            for (int x = 0; x < N; x++) {
                if (!(have >> x & 1)) continue;
                const uint32_t bx = (bases >> (2 * x)) & 3u;
                for (int y = x + 1; y < N; y++) if (have >> y & 1) acc += G.s[bx][(bases >> (2 * y)) & 3u];
            }
        }
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) out[i] = acc;
    }
}

int chain_device_graph(mauve_ctx *c, int N, int64_t maxlen_in, const uint32_t *seg0, uint32_t nseg, ChainGraphHost *G, bool graph_to_host, const mauve_scoring *sp_scoring)
{
    static const bool trace = getenv("MAUVE_TRACE") != nullptr;
    const double t0 = now_ms();
    const int64_t nm = c->dev_rec_n;
    if (nm <= 0 || nm >= (1LL << 31)) { c->err = "chain_device: no device-resident match list"; return MAUVE_ERR_STATE; }
    const uint32_t n = (uint32_t)nm;
    int64_t maxlen = std::max<int64_t>(1, maxlen_in);
    int pos_bits = 1; while (pos_bits < 31 && (1LL << pos_bits) <= maxlen) pos_bits++;
    const uint32_t dead_key = 1u << pos_bits;                      // one above every left end
    HIPCHK(c, c->ch_len.ensure((size_t)n * 4 * 4));               // len, node_of, lcb, gap id
    HIPCHK(c, c->ch_st.ensure((size_t)n * N * 4));
    HIPCHK(c, c->ch_crop.ensure((size_t)n * 8));
    const uint32_t nb = (n + CH_TILE - 1) / CH_TILE;
    HIPCHK(c, c->ch_ent.ensure((size_t)n * 4 * 6 + 64 + (size_t)n + 64 + (size_t)nb * 4 * (2 + (size_t)N)));   // 2 x (key, val) for the sort, the cluster starts, flags, tile aggregates
    HIPCHK(c, c->ch_ord.ensure((size_t)n * N * 4 * 3));           // ord[N][n], ordc[N][n], okey[N][n]: the genomes' orders and their keys as the cluster passes leave them
    HIPCHK(c, c->ch_rank.ensure((size_t)n * N * 4));
    HIPCHK(c, c->ch_cnt.ensure(256));
    if (seg0) HIPCHK(c, c->ch_big.ensure((size_t)n * 29 + 64));   // working arrays of the big clusters
    int32_t *len = c->ch_len.as<int32_t>(), *node_of = len + n;
    uint32_t *gapid = seg0 ? reinterpret_cast<uint32_t *>(len + 3 * (size_t)n) : nullptr;
    int32_t *st = c->ch_st.as<int32_t>();
    uint32_t *crop = c->ch_crop.as<uint32_t>();
    uint32_t *k1 = c->ch_ent.as<uint32_t>(), *v1 = k1 + n, *k2 = v1 + n, *v2 = k2 + n, *sl = v2 + n;     // sl: cluster starts (up to n + 1)
    uint8_t *cflag = reinterpret_cast<uint8_t *>(sl + 2 * (size_t)n + 16);
    uint32_t *bmax = reinterpret_cast<uint32_t *>(cflag + (((size_t)n + 63) & ~(size_t)63)), *balive = bmax + nb, *bcnt = balive + nb;
    uint32_t *ord = c->ch_ord.as<uint32_t>(), *ordc = ord + (size_t)n * N, *okey = ordc + (size_t)n * N;
    uint32_t *rank = c->ch_rank.as<uint32_t>();
    uint32_t *cnt = c->ch_cnt.as<uint32_t>();                     // [0] na, [1] K, [2] link check, [3] fail, [4] clusters, [5] alive entries of the genome in turn,
                                                                  // [8+g] entries of genome g alive before its cluster pass (the domain of FinalRanks)
    const int64_t *rlen = c->sorted_rec.as<int64_t>(), *rst = rlen + n;
    const uint32_t blocks = (n + 255) / 256;
    static const int big_max = []() { const char *e = getenv("MAUVE_CH_BIG_MAX"); const int v = e ? atoi(e) : 2048; return v < 1 ? 1 : v; }();        // cluster size the one-lane kernel still takes (tests lower it)
    static const int cl_max = []() { const char *e = getenv("MAUVE_CH_CL_MAX"); const int v = e ? atoi(e) : CH_CL_MAX; return v < 1 ? 1 : (v > CH_CL_MAX ? CH_CL_MAX : v); }();   // (tests lower it)
    // the graph arrays are sized for the worst case K = n
    HIPCHK(c, c->ch_graph.ensure((size_t)n * (8 + 4 + (size_t)N * 4 * 3 + 4) + 8 + (size_t)n * (8 + 4 + (size_t)N * 8) + 64));     // the arrays + their packed copy
    unsigned long long *weight = c->ch_graph.as<unsigned long long>();
    if (seg0) hipLaunchKernelGGL(ch_init_seg, dim3(blocks), dim3(256), 0, c->stream, rlen, rst, n, N, seg0, nseg, len, st, crop, gapid, cnt, weight);
    else hipLaunchKernelGGL(ch_init, dim3(blocks), dim3(256), 0, c->stream, rlen, rst, n, N, len, st, crop, cnt, weight);
    const int sort_passes = (pos_bits + 1 + 7) / 8;
    for (int g = 0; g < N; g++) {
        // The genome's order and keys stay where its cluster pass leaves them -- (okey, ord)[g] -- and FinalRanks compacts from there at the end: no
        // compaction of the survivors per genome (two launches each).  The radix sort ends in its alternate buffers after an odd number of passes, so
        // the buffers are handed to it the way that makes the result land there.
        uint32_t *og = ord + (size_t)g * n, *kg = okey + (size_t)g * n;
        // genome 0: the list is in canonical order, i.e. already ordered by its left ends there (equal left ends keep their list order: every pass
        // below breaks ties by the list index, as the host chain does), nobody is dead yet
        // (a recursion batch starts with its non-forward matches dead: sorted like any other genome)
        const bool sorted = g > 0 || seg0;
        const bool odd = sorted && (sort_passes & 1);
        uint32_t *kk = odd ? k1 : kg, *vv = odd ? v1 : og;
        hipLaunchKernelGGL(ch_keys, dim3(blocks), dim3(256), 0, c->stream, len, st, n, N, g, dead_key, kk, vv);
        if (sorted) { int rc = sort_pairs_u32(c, n, pos_bits + 1, &kk, &vv, odd ? kg : k2, odd ? og : v2, MAUVE_K_MISC); if (rc) return rc; }
        if (kk != kg || vv != og) {                               // (the sort took another number of passes than counted here: put the result in place)
            HIPCHK(c, hipMemcpyAsync(kg, kk, (size_t)n * 4, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(og, vv, (size_t)n * 4, hipMemcpyDeviceToDevice, c->stream));
            kk = kg; vv = og;
        }
        hipLaunchKernelGGL(cl_partial, dim3(nb), dim3(256), 0, c->stream, len, n, dead_key, kk, vv, bmax, balive);
        hipLaunchKernelGGL(cl_flags, dim3(nb), dim3(256), 0, c->stream, len, n, dead_key, kk, vv, bmax, balive, nb, cflag, cnt, bcnt, cnt + 8 + g);
        const ClusterStarts cs{cflag, n, sl, cnt};
        hipLaunchKernelGGL((cmp_write<ClusterStarts>), dim3(nb), dim3(256), 0, c->stream, cs, bcnt);
        hipLaunchKernelGGL(ch_cluster_pass, dim3(blocks), dim3(256), 0, c->stream, len, st, N, g, dead_key, kk, vv, sl, cnt + 4, cnt + 3, seg0 ? 1 : 0, cl_max);
        if (seg0) hipLaunchKernelGGL(ch_cluster_pass_big, dim3((n + 63) / 64), dim3(64), 0, c->stream, len, st, N, g, dead_key, kk, vv, sl, cnt + 4,
                                     c->ch_big.as<uint32_t>(), n, cl_max, cnt + 3, big_max);
    }
    const FinalRanks fr{len, n, ord, okey, dead_key, ordc, rank, cnt};
    hipLaunchKernelGGL((cmp_count<FinalRanks>), dim3(nb, N), dim3(256), 0, c->stream, fr, bcnt);
    hipLaunchKernelGGL((cmp_write<FinalRanks>), dim3(nb, N), dim3(256), 0, c->stream, fr, bcnt);
    uint32_t *orient = reinterpret_cast<uint32_t *>(weight + n);
    int32_t *prevv = reinterpret_cast<int32_t *>(orient + n), *nextv = prevv + (size_t)n * N, *seq = nextv + (size_t)n * N;
    int32_t *final_dev = seq + (size_t)n * N;
    const int64_t *mw = nullptr;
    if (sp_scoring) {                                         // score-weighted LCBs: the scores of the records as the elimination left them
        if (N > 16) { c->err = "sum-of-pairs LCB scoring: at most 16 genomes"; return MAUVE_ERR_LIMIT; }
        HIPCHK(c, c->ch_mw.ensure((size_t)n * 8 + 64));
        ChSpGenomes SG; memset(&SG, 0, sizeof SG);
        for (int g = 0; g < N; g++) SG.word_off[g] = c->word_off[(size_t)g];
        memcpy(SG.s, sp_scoring->matrix, sizeof SG.s);
        hipLaunchKernelGGL(ch_sp_scores, dim3((uint32_t)std::min<size_t>(((size_t)n + 3) / 4, 256 * 8)), dim3(256), 0, c->stream, c->genomes.as<uint64_t>(), SG, N, len, st, n,
                           c->ch_mw.as<int64_t>());
        mw = c->ch_mw.as<int64_t>();
    }
    const LcbNodes ln{len, st, n, N, ordc, rank, cnt, node_of, weight, orient, gapid, mw};
    hipLaunchKernelGGL((cmp_count<LcbNodes>), dim3(nb), dim3(256), 0, c->stream, ln, bcnt);
    hipLaunchKernelGGL((cmp_write<LcbNodes>), dim3(nb), dim3(256), 0, c->stream, ln, bcnt);
    const NodeSeq ns{n, ordc, cnt, node_of, seq};
    hipLaunchKernelGGL((cmp_count<NodeSeq>), dim3(nb, N), dim3(256), 0, c->stream, ns, bcnt);
    hipLaunchKernelGGL((cmp_write<NodeSeq>), dim3(nb, N), dim3(256), 0, c->stream, ns, bcnt);
    hipLaunchKernelGGL(ch_links, dim3(blocks, N), dim3(256), 0, c->stream, N, cnt, seq, prevv, nextv);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, c->pin_chain.ensure(256));
    HIPCHK(c, hipMemcpyAsync(c->pin_chain.p, cnt, 256, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const uint32_t *hc = c->pin_chain.as<uint32_t>();
    if (hc[3]) { c->err = "chain_device: overlap cluster beyond the per-thread limit"; return MAUVE_ERR_LIMIT; }      // caller falls back to the host chain
    const uint32_t na = hc[0], K = hc[1];
    G->na = na; G->K = K; G->weight = nullptr; G->orient = nullptr; G->prev = nullptr; G->next = nullptr; G->final_stage = nullptr; G->final_dev = final_dev;
    if (K && graph_to_host) {
        // graph to the host: weight[K], orient[K], prev[K*N], next[K*N]
        const size_t gbytes = (size_t)K * (8 + 4 + (size_t)N * 8) + 64;
        HIPCHK(c, c->pin_chain.ensure(256 + gbytes + (size_t)K * 4));
        char *pg = c->pin_chain.as<char>() + 256;
        int64_t *hw = reinterpret_cast<int64_t *>(pg);
        uint32_t *ho = reinterpret_cast<uint32_t *>(hw + K);
        int32_t *hp = reinterpret_cast<int32_t *>(ho + K), *hn = hp + (size_t)K * N;
        // packed on the device (the arrays are K-prefixes of capacity-n arrays), one copy
        const size_t pbytes = (size_t)K * (8 + 4 + (size_t)N * 8) + 4;
        unsigned long long *pack = reinterpret_cast<unsigned long long *>((reinterpret_cast<uintptr_t>(final_dev + n) + 7) & ~(uintptr_t)7);   // behind the graph arrays
        hipLaunchKernelGGL(ch_pack_graph, dim3((K * (uint32_t)N + 255) / 256), dim3(256), 0, c->stream, K, N, weight, orient, prevv, nextv, cnt, pack);
        HIPCHK(c, hipMemcpyAsync(pg, pack, pbytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (*reinterpret_cast<uint32_t *>(hn + (size_t)K * N)) { c->err = "chain_device: node lists inconsistent"; return MAUVE_ERR_LIMIT; }
        G->weight = hw; G->orient = ho; G->prev = hp; G->next = hn; G->final_stage = reinterpret_cast<int32_t *>(pg + gbytes);
    }
    if (trace) fprintf(stderr, "[trace] chain (device): eliminate+nodes+graph %.3f ms (na=%u K=%u)\n", now_ms() - t0, na, K);
    return MAUVE_OK;
}

// Elimination, LCB graph, greedy breakpoint elimination, labels: everything stays on the device (cropped records in
// c->ch_len / c->ch_st, final LCB id per match behind them).  chain_device_copy_back brings them to the host.
int chain_device_core(mauve_ctx *c, int N, int64_t min_weight, bool collinear, int64_t &n_lcb, const mauve_scoring *sp_scoring)
{
    static const bool trace = getenv("MAUVE_TRACE") != nullptr;
    int64_t maxlen = 1; for (int g = 0; g < N; g++) maxlen = std::max<int64_t>(maxlen, c->lens[(size_t)g]);
    ChainGraphHost G;
    int rc = chain_device_graph(c, N, maxlen, nullptr, 0, &G, true, sp_scoring);
    if (rc) return rc;
    const double t1 = now_ms();
    const uint32_t n = (uint32_t)c->dev_rec_n, K = G.K, blocks = (n + 255) / 256;
    int32_t *len = c->ch_len.as<int32_t>(), *node_of = len + n, *lcb = node_of + n;
    n_lcb = 0;
    if (K) {
        std::vector<int64_t> final_id;
        lcb_greedy(N, (int32_t)K, G.weight, G.orient, G.prev, G.next, min_weight, collinear, final_id, n_lcb);
        for (uint32_t i = 0; i < K; i++) G.final_stage[i] = (int32_t)final_id[i];
        HIPCHK(c, hipMemcpyAsync(G.final_dev, G.final_stage, (size_t)K * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(ch_label, dim3(blocks), dim3(256), 0, c->stream, len, n, node_of, G.final_dev, lcb);
        HIPCHK(c, hipGetLastError());
    } else {
        HIPCHK(c, hipMemsetAsync(lcb, 0xff, (size_t)n * 4, c->stream));
    }
    if (trace) fprintf(stderr, "[trace] chain (device): greedy+labels %.3f ms\n", now_ms() - t1);
    return MAUVE_OK;
}

// A recursion batch chained on the device (recursive.cpp): the N-way forward matches of ALL gaps of the batch as one
// list on the virtual genomes.  Overlap elimination of the whole list equals the elimination gap by gap (matches of
// different gaps lie in different segments of every virtual genome, so they never share an overlap cluster), the LCB
// nodes are confined to their gaps, and the collinear rule (one collinear chain per gap) runs on the host
// per gap over the compact graph -- nearly every gap has a single node.  Out: cropped records (hl, hs: int32, list
// order) and survive[i] for every match of the list.
int chain_device_gaps(mauve_ctx *c, int N, int64_t maxlen, const uint32_t *seg0_dev, uint32_t nseg, const int32_t **hl_out, const int32_t **hs_out,
                      std::vector<uint8_t> &survive)
{
    ChainGraphHost G;
    {   // page-locked room for the graph (at most one node per match) AND the records behind it, before the graph lands there
        const size_t n0 = (size_t)c->dev_rec_n;
        HIPCHK(c, c->pin_chain.ensure(256 + n0 * (8 + 4 + (size_t)N * 8) + 64 + n0 * 4 + 64 + n0 * 4 * (3 + (size_t)N) + 64));
    }
    int rc = chain_device_graph(c, N, maxlen, seg0_dev, nseg, &G);
    if (rc) return rc;
    const uint32_t n = (uint32_t)c->dev_rec_n, K = G.K;
    const int32_t *len = c->ch_len.as<int32_t>(), *node_of = len + n;
    const uint32_t *gapid = reinterpret_cast<const uint32_t *>(len + 3 * (size_t)n);
    const int32_t *st = c->ch_st.as<int32_t>();
    // the graph is consumed before pin_chain is reused for the records
    std::vector<uint8_t> node_ok((size_t)K, 1);
    std::vector<int64_t> w; std::vector<uint32_t> o; std::vector<int32_t> pv, nx; std::vector<int64_t> final_id;
    // records, nodes and gap ids of the matches
    std::vector<int32_t> g_of_node((size_t)K, -1);
    {
        // node -> gap needs the per-match arrays: fetch them behind the graph block
        const size_t gbytes = (size_t)K * (8 + 4 + (size_t)N * 8) + 64 + (size_t)K * 4;
        char *pg = c->pin_chain.as<char>() + 256;
        int32_t *hl = reinterpret_cast<int32_t *>(pg + ((gbytes + 63) & ~(size_t)63)), *hs = hl + n, *hnode = hs + (size_t)n * N;
        uint32_t *hgap = reinterpret_cast<uint32_t *>(hnode + n);
        HIPCHK(c, hipMemcpyAsync(hl, len, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(hs, st, (size_t)n * N * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(hnode, node_of, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(hgap, gapid, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (uint32_t i = 0; i < n; i++) if (hl[i] > 0) g_of_node[(size_t)hnode[i]] = (int32_t)hgap[i];
        // per gap: its nodes are consecutive (genome-0 order); more than one -> the collinear rule on the sub-graph
        for (uint32_t n0 = 0; n0 < K;) {
            uint32_t n1 = n0 + 1;
            while (n1 < K && g_of_node[n1] == g_of_node[n0]) n1++;
            const int32_t kk = (int32_t)(n1 - n0);
            if (kk > 1) {
                w.assign(G.weight + n0, G.weight + n1); o.assign(G.orient + n0, G.orient + n1);
                pv.resize((size_t)kk * N); nx.resize((size_t)kk * N);
                for (int32_t j = 0; j < kk; j++)
                    for (int g = 0; g < N; g++) {
                        const int32_t p = G.prev[(size_t)(n0 + j) * N + g], q = G.next[(size_t)(n0 + j) * N + g];
                        pv[(size_t)j * N + g] = (p >= (int32_t)n0 && p < (int32_t)n1) ? p - (int32_t)n0 : -1;
                        nx[(size_t)j * N + g] = (q >= (int32_t)n0 && q < (int32_t)n1) ? q - (int32_t)n0 : -1;
                    }
                int64_t nl = 0;
                lcb_greedy(N, kk, w.data(), o.data(), pv.data(), nx.data(), 0, true, final_id, nl);
                for (int32_t j = 0; j < kk; j++) node_ok[(size_t)n0 + j] = final_id[(size_t)j] >= 0;
            }
            n0 = n1;
        }
        survive.assign((size_t)n, 0);
        for (uint32_t i = 0; i < n; i++) survive[i] = hl[i] > 0 && node_ok[(size_t)hnode[i]];
        *hl_out = hl; *hs_out = hs;
    }
    return MAUVE_OK;
}


// The same with the collinear rule on the device as well (gap_greedy) and only the SURVIVORS coming back: cropped records
// (hl, hs: int32) and the gap of each (hgap), in list order; *ns_out of them.  MAUVE_ERR_LIMIT: a gap beyond the kernel's LDS
// slice, or an overlap cluster beyond the per-thread limit -- the caller takes the host route for the batch.
int chain_device_gaps_compact(mauve_ctx *c, int N, int64_t maxlen, const uint32_t *seg0_dev, uint32_t nseg, const int32_t **hl_out, const int32_t **hs_out,
                              const uint32_t **hgap_out, uint32_t *ns_out)
{
    static const bool trace = getenv("MAUVE_TRACE") != nullptr;
    ChainGraphHost G;
    int rc = chain_device_graph(c, N, maxlen, seg0_dev, nseg, &G, false);
    if (rc) return rc;
    const double t0 = now_ms();
    const uint32_t n = (uint32_t)c->dev_rec_n, K = G.K, blocks = (n + 255) / 256, nb = (n + TILE - 1) / TILE;
    *ns_out = 0;
    if (K == 0) return MAUVE_OK;
    const int32_t *len = c->ch_len.as<int32_t>(), *node_of = len + n;
    const uint32_t *gapid = reinterpret_cast<const uint32_t *>(len + 3 * (size_t)n);
    const int32_t *st = c->ch_st.as<int32_t>();
    const unsigned long long *weight = c->ch_graph.as<unsigned long long>();
    const uint32_t *orient = reinterpret_cast<const uint32_t *>(weight + n);
    const int32_t *prevv = reinterpret_cast<const int32_t *>(orient + n), *nextv = prevv + (size_t)n * N;
    // work area: node_gap[K], ranges[2K], counters, node_ok[K], tile counts, compact records
    auto up8 = [](size_t x) { return (x + 7) & ~(size_t)7; };
    const size_t o_rng = (size_t)K * 4, o_cnt = o_rng + (size_t)K * 8, o_ok = o_cnt + 64, o_bc = up8(o_ok + K), o_len = o_bc + (size_t)nb * 4 + 64,
                 o_st = o_len + (size_t)n * 4, o_gap = o_st + (size_t)n * N * 4, total = o_gap + (size_t)n * 4;
    HIPCHK(c, c->gap_work.ensure(total + 64));
    char *wk = c->gap_work.as<char>();
    uint32_t *node_gap = reinterpret_cast<uint32_t *>(wk), *ranges = reinterpret_cast<uint32_t *>(wk + o_rng), *cnt = reinterpret_cast<uint32_t *>(wk + o_cnt);
    uint8_t *node_ok = reinterpret_cast<uint8_t *>(wk + o_ok);
    uint32_t *bcnt = reinterpret_cast<uint32_t *>(wk + o_bc);
    int32_t *olen = reinterpret_cast<int32_t *>(wk + o_len), *ost = reinterpret_cast<int32_t *>(wk + o_st);
    uint32_t *ogap = reinterpret_cast<uint32_t *>(wk + o_gap);
    HIPCHK(c, hipMemsetAsync(cnt, 0, 64, c->stream));
    hipLaunchKernelGGL(gap_node_gap, dim3(blocks), dim3(256), 0, c->stream, len, node_of, gapid, n, node_gap);
    hipLaunchKernelGGL(gap_ranges, dim3((K + 255) / 256), dim3(256), 0, c->stream, node_gap, K, node_ok, ranges, cnt);
    // (the number of ranges is a device value: the greedy launch is sized for the worst case and strides over what there is)
    hipLaunchKernelGGL(gap_greedy, dim3(std::min<uint32_t>((K + 1) / 2, 256 * 16)), dim3(64), 0, c->stream, ranges, cnt, N, weight, prevv, nextv, node_ok, cnt + 1);
    const GapSurvivors gs{len, st, node_of, gapid, node_ok, n, N, olen, ost, ogap, cnt + 2};
    hipLaunchKernelGGL((cmp_count<GapSurvivors>), dim3(nb), dim3(256), 0, c->stream, gs, bcnt);
    hipLaunchKernelGGL((cmp_write<GapSurvivors>), dim3(nb), dim3(256), 0, c->stream, gs, bcnt);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, c->pin_chain.ensure(256));
    HIPCHK(c, hipMemcpyAsync(c->pin_chain.p, cnt, 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const uint32_t *hc = c->pin_chain.as<uint32_t>();
    if (hc[1]) { c->err = "chain_device_gaps: a gap beyond the device kernel's node limit"; return MAUVE_ERR_LIMIT; }
    const uint32_t ns = hc[2], nr = hc[0];
    HIPCHK(c, c->pin_chain.ensure(256 + (size_t)ns * 4 * (2 + (size_t)N) + 64));
    int32_t *hl = reinterpret_cast<int32_t *>(c->pin_chain.as<char>() + 256), *hs = hl + ns;
    uint32_t *hg = reinterpret_cast<uint32_t *>(hs + (size_t)ns * N);
    if (ns) {
        HIPCHK(c, hipMemcpyAsync(hl, olen, (size_t)ns * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(hs, ost, (size_t)ns * N * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(hg, ogap, (size_t)ns * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    *hl_out = hl; *hs_out = hs; *hgap_out = hg; *ns_out = ns;
    if (trace) fprintf(stderr, "[trace] chain (device): collinear rule on %u gaps with several nodes, %u of %u matches survive, %.3f ms\n", nr, ns, n, now_ms() - t0);
    return MAUVE_OK;
}

// the cropped list and its labels back to the host (dead records stay in the list with length 0 / LCB -1)
int chain_device_copy_back(mauve_ctx *c, int N, MatchVec &m, std::vector<int64_t> &match_lcb)
{
    static const bool trace = getenv("MAUVE_TRACE") != nullptr;
    const double t2 = now_ms();
    const uint32_t n = (uint32_t)c->dev_rec_n;
    const int32_t *len = c->ch_len.as<int32_t>(), *lcb = len + 2 * (size_t)n;
    const int32_t *st = c->ch_st.as<int32_t>();
    match_lcb.assign((size_t)n, -1);
    const size_t rb = (size_t)n * 4 * (2 + (size_t)N);
    HIPCHK(c, c->pin_chain.ensure(256 + rb));
    int32_t *hl = reinterpret_cast<int32_t *>(c->pin_chain.as<char>() + 256), *hs = hl + n, *hb = hs + (size_t)n * N;
    HIPCHK(c, hipMemcpyAsync(hl, len, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(hs, st, (size_t)n * N * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(hb, lcb, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    m.N = N; m.resize((size_t)n);
    for (uint32_t i = 0; i < n; i++) {
        int64_t *r = &m.d[(size_t)i * (1 + N)];
        if (hl[i] > 0) {                                           // a dead record keeps what it was when it died on the host path too;
            r[0] = hl[i];                                          // nothing downstream reads it (LCB -1)
            for (int g = 0; g < N; g++) r[1 + g] = hs[(size_t)i * N + g];
        } else {
            r[0] = c->match_len[(size_t)i];
            for (int g = 0; g < N; g++) r[1 + g] = c->match_start[(size_t)i * N + g];
        }
        match_lcb[i] = hb[i];
    }
    {   // the survivors in canonical order of their cropped records, as host_eliminate_overlaps leaves them; dead ones last
        int64_t prev = 0; bool sorted = true;
        for (uint32_t i = 0; i < n && sorted; i++) { if (hl[i] <= 0) continue; const int64_t s0 = std::llabs(m.st(i)[0]); sorted = s0 > prev; prev = s0; }
        if (!sorted) {
            std::vector<uint32_t> perm; perm.reserve(n);
            for (uint32_t i = 0; i < n; i++) if (hl[i] > 0) perm.push_back(i);
            std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return std::llabs(m.st(a)[0]) < std::llabs(m.st(b)[0]); });
            for (uint32_t i = 0; i < n; i++) if (hl[i] <= 0) perm.push_back(i);
            MatchVec t(N); t.d.resize(m.d.size()); std::vector<int64_t> tl((size_t)n);
            for (uint32_t k = 0; k < n; k++) { std::copy(m.rec(perm[k]), m.rec(perm[k]) + 1 + N, t.d.begin() + (std::ptrdiff_t)((size_t)k * (1 + N))); tl[k] = match_lcb[perm[k]]; }
            m.d.swap(t.d); match_lcb.swap(tl);
        }
    }
    if (trace) fprintf(stderr, "[trace] chain (device): copy back %.3f ms\n", now_ms() - t2);
    return MAUVE_OK;
}

int chain_device(mauve_ctx *c, int N, int64_t min_weight, bool collinear, MatchVec &m, std::vector<int64_t> &match_lcb, int64_t &n_lcb, const mauve_scoring *sp_scoring)
{
    int rc = chain_device_core(c, N, min_weight, collinear, n_lcb, sp_scoring);
    if (rc) return rc;
    return chain_device_copy_back(c, N, m, match_lcb);
}
