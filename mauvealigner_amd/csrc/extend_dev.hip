// extend_dev.hip -- LCB extension (lcb_extension, mauveAligner.cpp:95; Aligner::SetMaxExtensionIterations :687-690; frozen
// form DESIGN.md S10) without taking the chains off the device.
//
// A round searches N-way matches where no LCB lies and keeps them if they add anchored columns.  What it needs of the
// chains is tiny: the LCBs' extents (to know where to search) and their weights (to re-run the greedy elimination).  The
// new matches lie outside every LCB extent in every genome, so they cannot come between two matches of an LCB: the LCB
// graph of "survivors + new matches" is the graph whose nodes are the old LCBs and the new matches -- a hundred nodes, not
// fifty thousand matches -- and host_lcb_chain on those units (with the LCB weights as match weights) is exactly the
// recomputation the rule prescribes.  So per round:
//   host   : pieces = complement of the LCB extents per genome (>= one seed long)
//   device : gather the pieces into small virtual genomes (a flagged separator base between two pieces: no window, and no
//            run of agreeing windows, crosses from one piece into the next), one ordinary N-way seed pass over them
//            (a few 10^4 windows instead of a masked pass over all 2.5 * 10^7)
//   host   : map the handful of matches back, eliminate overlaps among them, re-chain the units, update the LCB table
// and after the last round one kernel slips the kept matches into the device-resident anchor list (both are ordered by
// their start in genome 0) and recounts the gaps the recursion would look at.  The 52 k anchors never move.
#include "common.hpp"
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace {

// first / last anchor of every LCB (anchors are in chain order: LCB by LCB)
__global__ void __launch_bounds__(256) ext_lcb_ends(const int32_t *__restrict__ alcb, uint32_t na, uint32_t *__restrict__ first_a, uint32_t *__restrict__ last_a)
{
    const uint32_t a = blockIdx.x * 256u + threadIdx.x;
    if (a >= na) return;
    const int32_t l = alcb[a];
    if (a == 0 || alcb[a - 1] != l) first_a[l] = a;
    if (a + 1 == na || alcb[a + 1] != l) last_a[l] = a;
}

// signed extents of every LCB in every genome (negative: the LCB is reverse there), from its first and last anchor
__global__ void __launch_bounds__(256) ext_lcb_extents(const int32_t *__restrict__ alen, const int32_t *__restrict__ ast, int N, uint32_t nl,
                                                       const uint32_t *__restrict__ first_a, const uint32_t *__restrict__ last_a,
                                                       int64_t *__restrict__ left, int64_t *__restrict__ right)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= nl * (uint32_t)N) return;
    const uint32_t l = t / (uint32_t)N; const int g = (int)(t % (uint32_t)N);
    const uint32_t a0 = first_a[l], a1 = last_a[l];
    const int64_t s0 = ast[(size_t)a0 * N + g], s1 = ast[(size_t)a1 * N + g];
    int64_t le, re;
    if (s0 > 0) { le = s0; re = s1 + alen[a1] - 1; }
    else { le = -s1; re = -s0 + alen[a0] - 1; }
    left[t] = s0 < 0 ? -le : le; right[t] = s0 < 0 ? -re : re;
}

// Virtual genomes of one round: genome g = its valid pieces one after the other, a separator base (code 0, flagged in the
// round's bitmap) between two of them.  Thread (g, j) writes packed word j.  piece k of genome g: virtual start vs[g*(K+1)+k]
// (vs[..+K] = total length), real 0-based start rs[g*K+k], length ln[g*K+k]; unused entries have length 0 at the total.
struct ExtGatherArgs { uint64_t src_word_off[MAUVE_MAX_SEQ], dst_word_off[MAUVE_MAX_SEQ], dst_words[MAUVE_MAX_SEQ]; };
// (one thread per BASE -- one search of the piece table, one load -- and the wave packs its 64 bases into two words by ballots; a thread per word that
// walked its 32 bases one after the other, three dependent loads each, took 20 us for a few kilobases)
__device__ __forceinline__ uint64_t spread_bits32(uint32_t x)       // bit b -> bit 2 b
{
    uint64_t v = x;
    v = (v | (v << 16)) & 0x0000ffff0000ffffULL; v = (v | (v << 8)) & 0x00ff00ff00ff00ffULL; v = (v | (v << 4)) & 0x0f0f0f0f0f0f0f0fULL;
    v = (v | (v << 2)) & 0x3333333333333333ULL; v = (v | (v << 1)) & 0x5555555555555555ULL;
    return v;
}
__global__ void __launch_bounds__(256) ext_gather(const uint64_t *__restrict__ genomes, uint64_t *__restrict__ out, ExtGatherArgs ga,
                                                  const uint32_t *__restrict__ vs, const int64_t *__restrict__ rs, const uint32_t *__restrict__ ln, uint32_t K)
{
    const int g = blockIdx.y;
    const uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x;      // virtual base; a wave covers words 2 w, 2 w + 1
    const uint64_t w0 = p >> 5 & ~(uint64_t)1;
    if (w0 >= ga.dst_words[g]) return;                                 // (wave-uniform)
    const uint32_t *v = vs + (size_t)g * (K + 1);
    const uint32_t tot = v[K];
    uint32_t code = 0;
    if (p < tot) {
        uint32_t lo = 0, hi = K;                                      // last k with v[k] <= p
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (v[mid] <= p) lo = mid; else hi = mid; }
        const uint64_t off = p - v[lo];
        if (off < ln[(size_t)g * K + lo]) {                           // (else: the separator behind piece lo)
            const int64_t src = rs[(size_t)g * K + lo] + (int64_t)off;
            code = (uint32_t)((genomes[ga.src_word_off[g] + (src >> 5)] >> (2 * (src & 31))) & 3ULL);
        }
    }
    const uint64_t b0 = __ballot(code & 1u), b1 = __ballot(code & 2u);
    const int lane = threadIdx.x & 63;
    if (lane < 2 && w0 + lane < ga.dst_words[g]) {
        const uint32_t h0 = (uint32_t)(b0 >> (32 * lane)), h1 = (uint32_t)(b1 >> (32 * lane));
        out[ga.dst_word_off[g] + w0 + lane] = spread_bits32(h0) | (spread_bits32(h1) << 1);
    }
}

// the kept matches slipped into the anchor list: both are ordered by their start in genome 0 (always forward there)
__global__ void __launch_bounds__(256) ext_merge(const int32_t *__restrict__ alen, const int32_t *__restrict__ ast, const int32_t *__restrict__ alcb, uint32_t na, int N,
                                                 const int32_t *__restrict__ xlen, const int32_t *__restrict__ xst, const int32_t *__restrict__ xlcb, uint32_t nx,
                                                 const int32_t *__restrict__ lmap, int32_t *__restrict__ olen, int32_t *__restrict__ ost, int32_t *__restrict__ olcb)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j < na) {
        const int32_t s0 = ast[(size_t)j * N];
        uint32_t lo = 0, hi = nx;                                 // added matches that start before this anchor
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (xst[(size_t)mid * N] < s0) lo = mid + 1; else hi = mid; }
        const uint32_t o = j + lo;
        olen[o] = alen[j]; olcb[o] = lmap[alcb[j]];
        for (int g = 0; g < N; g++) ost[(size_t)o * N + g] = ast[(size_t)j * N + g];
    } else if (j < na + nx) {
        const uint32_t t = j - na;
        const int32_t s0 = xst[(size_t)t * N];
        uint32_t lo = 0, hi = na;                                 // anchors that start before this match
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (ast[(size_t)mid * N] < s0) lo = mid + 1; else hi = mid; }
        const uint32_t o = lo + t;
        olen[o] = xlen[t]; olcb[o] = xlcb[t];
        for (int g = 0; g < N; g++) ost[(size_t)o * N + g] = xst[(size_t)t * N + g];
    }
}

// inter-anchor gaps the recursion would look at (longest side above min_gap; as co_gather counts them)
__global__ void __launch_bounds__(256) ext_count_rec(const int32_t *__restrict__ alen, const int32_t *__restrict__ ast, const int32_t *__restrict__ alcb, uint32_t na, int N,
                                                     int64_t min_gap, uint32_t *__restrict__ out)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j + 1 >= na || alcb[j] != alcb[j + 1]) return;
    int64_t mx = 0;
    for (int g = 0; g < N; g++) {
        const int64_t sa = ast[(size_t)j * N + g], sb = ast[(size_t)(j + 1) * N + g];
        int64_t lo, hi;
        if (sa > 0) { lo = sa + alen[j]; hi = sb - 1; } else { lo = -sb + alen[j + 1]; hi = -sa - 1; }
        mx = max(mx, hi - lo + 1);
    }
    if (mx > min_gap) atomicAdd(out, 1u);
}

// ... and which ones: flag[j] = 1 when the gap behind anchor j (same LCB) is one the recursion has to look at (the host's work list then
// visits those few thousand instead of testing every anchor pair)
__global__ void __launch_bounds__(256) ext_flag_rec(const int32_t *__restrict__ alen, const int32_t *__restrict__ ast, const int32_t *__restrict__ alcb, uint32_t na, int N,
                                                    int64_t min_gap, uint8_t *__restrict__ flag)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= na) return;
    uint8_t f = 0;
    if (j + 1 < na && alcb[j] == alcb[j + 1]) {
        int64_t mx = 0;
        for (int g = 0; g < N; g++) {
            const int64_t sa = ast[(size_t)j * N + g], sb = ast[(size_t)(j + 1) * N + g];
            int64_t lo, hi;
            if (sa > 0) { lo = sa + alen[j]; hi = sb - 1; } else { lo = -sb + alen[j + 1]; hi = -sa - 1; }
            mx = max(mx, hi - lo + 1);
        }
        f = mx > min_gap;
    }
    flag[j] = f;
}

struct LcbTab {                                                   // the LCBs as the rounds see them
    int N = 0; int64_t n = 0;
    std::vector<int64_t> lo, hi, weight;                          // [n * N] absolute extents, [n]
    std::vector<uint32_t> rev;                                    // bit g: reverse in genome g
};

}  // namespace

// the recursion's candidate gaps of a device-resident anchor list (chain order), as one byte per anchor in d_flag
int rec_gap_flags_device(mauve_ctx *c, const int32_t *alen, const int32_t *ast, const int32_t *alcb, int64_t na, int N, int64_t min_gap, uint8_t *d_flag)
{
    if (na <= 0) return MAUVE_OK;
    hipLaunchKernelGGL(ext_flag_rec, dim3((uint32_t)((na + 255) / 256)), dim3(256), 0, c->stream, alen, ast, alcb, (uint32_t)na, N, min_gap, d_flag);
    HIPCHK(c, hipGetLastError());
    return MAUVE_OK;
}

// anchors: na records in chain order at (*alen, *ast, *alcb) on the device, nl LCBs with weights in c->ch_lw.  On return the
// pointers name the extended list (c->ch_anch2 when anything was added), na / nl / n_rec are updated and lcb_weight holds the
// weights of the final LCBs.  Caller guarantees: no ambiguity / contig bitmaps, length-weighted LCBs.
int extend_lcbs_device(mauve_ctx *c, const mauve_params *p, int w, int64_t lcbw, int N, const int32_t **alen_io, const int32_t **ast_io,
                       const int32_t **alcb_io, int64_t *na_io, int64_t *nl_io, int64_t *n_rec_io, std::vector<int64_t> &lcb_weight)
{
    static const bool trace = getenv("MAUVE_TRACE") != nullptr;
    const double t0 = now_ms();
    const uint32_t na = (uint32_t)*na_io; const int64_t nl0 = *nl_io;
    const int32_t *alen = *alen_io, *ast = *ast_io, *alcb = *alcb_io;
    const uint32_t full = N >= 32 ? 0xffffffffu : ((1u << N) - 1);
    // ---- the LCB table: extents from the first / last anchor of every LCB, weights as chain_order_device summed them ----
    LcbTab T; T.N = N; T.n = nl0;
    {
        const size_t o_last = (size_t)nl0 * 4, o_left = (o_last + (size_t)nl0 * 4 + 7) & ~(size_t)7, o_right = o_left + (size_t)nl0 * N * 8, total = o_right + (size_t)nl0 * N * 8;
        HIPCHK(c, c->ext_work.ensure(total + 64));
        char *wk = c->ext_work.as<char>();
        uint32_t *first_a = reinterpret_cast<uint32_t *>(wk), *last_a = reinterpret_cast<uint32_t *>(wk + o_last);
        int64_t *left = reinterpret_cast<int64_t *>(wk + o_left), *right = reinterpret_cast<int64_t *>(wk + o_right);
        hipLaunchKernelGGL(ext_lcb_ends, dim3((na + 255) / 256), dim3(256), 0, c->stream, alcb, na, first_a, last_a);
        hipLaunchKernelGGL(ext_lcb_extents, dim3(((uint32_t)nl0 * N + 255) / 256), dim3(256), 0, c->stream, alen, ast, N, (uint32_t)nl0, first_a, last_a, left, right);
        HIPCHK(c, hipGetLastError());
        const size_t eb = 2 * (size_t)nl0 * N * 8;
        HIPCHK(c, c->pin_ext.ensure(eb + (size_t)nl0 * 8 + 64));
        HIPCHK(c, hipMemcpyAsync(c->pin_ext.p, left, eb, hipMemcpyDeviceToHost, c->stream));          // left and right are adjacent
        HIPCHK(c, hipMemcpyAsync(c->pin_ext.as<char>() + eb, c->ch_lw.p, (size_t)nl0 * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const int64_t *hl = c->pin_ext.as<int64_t>(), *hr = hl + (size_t)nl0 * N, *hw = hr + (size_t)nl0 * N;
        T.lo.resize((size_t)nl0 * N); T.hi.resize((size_t)nl0 * N); T.weight.assign(hw, hw + nl0); T.rev.assign((size_t)nl0, 0u);
        for (int64_t l = 0; l < nl0; l++)
            for (int g = 0; g < N; g++) {
                const int64_t a = hl[(size_t)l * N + g], b = hr[(size_t)l * N + g];
                T.lo[(size_t)l * N + g] = std::llabs(a); T.hi[(size_t)l * N + g] = std::llabs(b);
                if (a < 0) T.rev[(size_t)l] |= 1u << g;
            }
    }
    // the seed passes below reuse the seed workspace: a match list the main pass left on the device only (sorted_rec) is set aside
    const bool keep_pending = c->matches_pending; const int64_t keep_n = c->n_matches, keep_rec = c->dev_rec_n; const int keep_nseq = c->match_nseq;
    std::swap(c->sorted_rec, c->sorted_rec_keep);
    auto restore = [&]() {
        std::swap(c->sorted_rec, c->sorted_rec_keep);
        c->matches_pending = keep_pending; c->n_matches = keep_n; c->dev_rec_n = keep_rec; c->match_nseq = keep_nseq;
        if (keep_pending) { c->match_len.clear(); c->match_start.clear(); }
    };
    MatchVec added(N); std::vector<int64_t> added_lcb;            // kept matches (any order) and their LCB
    std::vector<int64_t> cur_of_orig((size_t)nl0);                // original LCB -> current LCB
    for (int64_t l = 0; l < nl0; l++) cur_of_orig[(size_t)l] = l;
    int w_e = w;
    for (int iter = 0; iter < p->max_extension_iters; iter++) {
        w_e -= 2;
        if (w_e < 5) break;
        const uint64_t pe = mauve_get_seed(w_e, 0);
        if (!pe) break;
        const int64_t span_e = mauve_seed_length(pe);
        const double tr0 = now_ms();
        // ---- pieces: per genome the complement of the LCB extents, at least one seed long ----
        std::vector<std::vector<std::pair<int64_t, int64_t>>> pieces((size_t)N);       // (1-based start, length)
        bool starved = false; size_t K = 0;
        std::vector<std::pair<int64_t, int64_t>> sp((size_t)T.n);
        for (int g = 0; g < N && !starved; g++) {
            for (int64_t l = 0; l < T.n; l++) sp[(size_t)l] = {T.lo[(size_t)l * N + g], T.hi[(size_t)l * N + g]};
            std::sort(sp.begin(), sp.end());
            int64_t cur = 1;
            for (int64_t l = 0; l <= T.n; l++) {
                const int64_t vlo = cur, vhi = l < T.n ? sp[(size_t)l].first - 1 : c->lens[(size_t)g];
                if (vhi - vlo + 1 >= span_e) pieces[(size_t)g].push_back({vlo, vhi - vlo + 1});
                if (l < T.n && sp[(size_t)l].second + 1 > cur) cur = sp[(size_t)l].second + 1;
            }
            if (pieces[(size_t)g].empty()) starved = true;
            K = std::max(K, pieces[(size_t)g].size());
        }
        if (starved) break;
        // ---- virtual genomes + separator bitmap ----
        GenomeSet vs; vs.buf = &c->rec_genomes; vs.nseq = N; vs.lens.assign((size_t)N, 0); vs.word_off.assign((size_t)N, 0); vs.mask_off.assign((size_t)N, 0);
        std::vector<uint32_t> vstart((size_t)N * (K + 1)), plen((size_t)N * K, 0u);
        std::vector<int64_t> rstart((size_t)N * K, 0);
        ExtGatherArgs ga; memset(&ga, 0, sizeof ga);
        size_t words = 0, mwords = 0; uint64_t max_words = 0;
        for (int g = 0; g < N; g++) {
            int64_t tot = 0;
            const auto &pg = pieces[(size_t)g];
            for (size_t k = 0; k < K; k++) {
                vstart[(size_t)g * (K + 1) + k] = (uint32_t)tot;
                if (k < pg.size()) { rstart[(size_t)g * K + k] = pg[k].first - 1; plen[(size_t)g * K + k] = (uint32_t)pg[k].second; tot += pg[k].second + (k + 1 < pg.size() ? 1 : 0); }
            }
            vstart[(size_t)g * (K + 1) + K] = (uint32_t)tot;
            vs.lens[(size_t)g] = tot;
            const size_t nw = mauve_packed_words(tot);
            vs.word_off[(size_t)g] = words; ga.src_word_off[g] = c->word_off[(size_t)g]; ga.dst_word_off[g] = words; ga.dst_words[g] = nw;
            max_words = std::max<uint64_t>(max_words, nw); words += nw;
            vs.mask_off[(size_t)g] = mwords; mwords += (size_t)((tot + 63) / 64) + 2;
        }
        HIPCHK(c, c->rec_genomes.ensure((words + 4) * sizeof(uint64_t)));
        const size_t b_vs = (vstart.size() * 4 + 7) & ~(size_t)7, b_rs = rstart.size() * 8, b_ln = (plen.size() * 4 + 7) & ~(size_t)7, b_mask = mwords * 8;
        HIPCHK(c, c->pin_ext.ensure(b_vs + b_rs + b_ln + b_mask + 64));
        HIPCHK(c, c->rec_vinv.ensure(b_mask + b_vs + b_rs + b_ln + 64));          // the bitmap, the side tables behind it: one upload
        char *pin = c->pin_ext.as<char>();
        uint64_t *bits = reinterpret_cast<uint64_t *>(pin);
        { char *ps = pin + b_mask; memcpy(ps, vstart.data(), vstart.size() * 4); memcpy(ps + b_vs, rstart.data(), b_rs); memcpy(ps + b_vs + b_rs, plen.data(), plen.size() * 4); }
        memset(bits, 0, b_mask);
        for (int g = 0; g < N; g++) {
            const auto &pg = pieces[(size_t)g];
            for (size_t k = 0; k + 1 < pg.size(); k++) {
                const uint64_t sep = (uint64_t)vstart[(size_t)g * (K + 1) + k] + (uint64_t)pg[k].second;       // the base behind piece k
                bits[vs.mask_off[(size_t)g] + (sep >> 6)] |= 1ULL << (sep & 63);
            }
        }
        char *side = c->rec_vinv.as<char>() + b_mask;
        HIPCHK(c, hipMemcpyAsync(c->rec_vinv.p, bits, b_mask + b_vs + b_rs + b_ln, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(ext_gather, dim3((uint32_t)((max_words * 32 + 255) / 256), (uint32_t)N), dim3(256), 0, c->stream, c->genomes.as<uint64_t>(),
                           c->rec_genomes.as<uint64_t>(), ga, (const uint32_t *)side, (const int64_t *)(side + b_vs), (const uint32_t *)(side + b_vs + b_rs), (uint32_t)K);
        HIPCHK(c, hipGetLastError());
        vs.vmask = &c->rec_vinv;
        int64_t nx = 0;
        const double tr1 = now_ms();
        int rc = seedpass_run(c, vs, pe, MAUVE_MODE_MEM, full, 1, nullptr, 0, &nx);      // (syncs the stream before the staging block is reused)
        if (rc) { restore(); return rc; }
        if (trace) fprintf(stderr, "[trace] lcb extension (device) round %d: pieces %.3f ms (%lld bases of genome 0), seed pass %.3f ms, %lld new matches\n", iter, tr1 - tr0,
                           (long long)vs.lens[0], now_ms() - tr1, (long long)nx);
        if (nx == 0) continue;
        const double tr2 = now_ms();
        // ---- back to real coordinates (the pieces are forward copies: a match keeps its strand signs) ----
        MatchVec ext(N); ext.resize((size_t)nx);
        for (int64_t i = 0; i < nx; i++) {
            ext.len((size_t)i) = c->match_len[(size_t)i];
            for (int g = 0; g < N; g++) {
                const int64_t s = c->match_start[(size_t)i * N + g], v0 = std::llabs(s) - 1;
                const uint32_t *vg = &vstart[(size_t)g * (K + 1)];
                const size_t k = (size_t)(std::upper_bound(vg, vg + K, (uint32_t)v0) - vg) - 1;
                const int64_t real = rstart[(size_t)g * K + k] + (v0 - vg[k]) + 1;
                ext.st((size_t)i)[g] = s < 0 ? -real : real;
            }
        }
        {   // canonical order (N-way records: |start 0|, starts, length), then the elimination among the new matches
            std::vector<size_t> idx(ext.size());
            for (size_t i = 0; i < idx.size(); i++) idx[i] = i;
            std::sort(idx.begin(), idx.end(), [&](size_t x, size_t y) {
                const int64_t *a = ext.rec(x), *b = ext.rec(y);
                const int64_t sa = std::llabs(a[1]), sb = std::llabs(b[1]);
                if (sa != sb) return sa < sb;
                for (int g = 0; g < N; g++) if (a[1 + g] != b[1 + g]) return a[1 + g] < b[1 + g];
                return a[0] < b[0];
            });
            MatchVec t(N); for (size_t i : idx) t.push(ext.rec(i));
            ext.d.swap(t.d);
        }
        host_eliminate_overlaps(ext);
        // ---- re-chain the units: old LCBs (weights as match weights) + new matches ----
        const size_t nu = (size_t)T.n + ext.size();
        MatchVec U(N); U.resize(nu); std::vector<int64_t> uw(nu), ul; int64_t nl2 = 0;
        for (int64_t l = 0; l < T.n; l++) {
            U.len((size_t)l) = 1; uw[(size_t)l] = T.weight[(size_t)l];
            for (int g = 0; g < N; g++) U.st((size_t)l)[g] = (T.rev[(size_t)l] >> g & 1u) ? -T.lo[(size_t)l * N + g] : T.lo[(size_t)l * N + g];
        }
        for (size_t i = 0; i < ext.size(); i++) {
            std::copy(ext.rec(i), ext.rec(i) + 1 + N, &U.d[((size_t)T.n + i) * (1 + N)]);
            uw[(size_t)T.n + i] = ext.len(i) * N;
        }
        host_lcb_chain(U, lcbw, p->collinear != 0, ul, nl2, nullptr, uw.data());
        static const bool force_decline = getenv("MAUVE_EXT_DECLINE") != nullptr;      // test knob: hand the first round that finds matches back as if an LCB had died
        for (int64_t l = 0; l < T.n; l++)
            if (ul[(size_t)l] < 0 || force_decline) {
                // an old LCB died in the re-chaining of the units (--collinear, which goes down to ONE LCB, is kept off this route; anything else that
                // gets here is handed back the same way): the caller takes the match-level rounds on the host, which make no such assumption
                restore(); c->err = "lcb extension: an LCB fell below the minimum weight while being extended"; return MAUVE_ERR_LIMIT;
            }
        bool grew = false;
        for (size_t i = 0; i < ext.size() && !grew; i++) grew = ul[(size_t)T.n + i] >= 0;
        if (trace) fprintf(stderr, "[trace] lcb extension (device) round %d: units %.3f ms, %lld -> %lld LCBs, %s\n", iter, now_ms() - tr2, (long long)T.n, (long long)nl2,
                           grew ? "kept" : "dropped");
        if (!grew) continue;                                      // the round is kept only if it raises the number of anchored columns
        LcbTab T2; T2.N = N; T2.n = nl2;
        T2.lo.assign((size_t)nl2 * N, 0); T2.hi.assign((size_t)nl2 * N, 0); T2.weight.assign((size_t)nl2, 0); T2.rev.assign((size_t)nl2, 0u);
        std::vector<uint8_t> seen((size_t)nl2, 0);
        for (size_t u = 0; u < nu; u++) {
            const int64_t l = ul[u]; if (l < 0) continue;         // (an old LCB never dies: its weight is above the minimum already)
            uint32_t rv = 0;
            for (int g = 0; g < N; g++) {
                int64_t a, b;
                if (u < (size_t)T.n) { a = T.lo[u * N + g]; b = T.hi[u * N + g]; if (T.rev[u] >> g & 1u) rv |= 1u << g; }
                else { const int64_t s = ext.st(u - (size_t)T.n)[g]; a = std::llabs(s); b = a + ext.len(u - (size_t)T.n) - 1; if (s < 0) rv |= 1u << g; }
                int64_t &L = T2.lo[(size_t)l * N + g], &H = T2.hi[(size_t)l * N + g];
                if (!seen[(size_t)l] || a < L) L = a;
                if (!seen[(size_t)l] || b > H) H = b;
            }
            T2.rev[(size_t)l] = rv; seen[(size_t)l] = 1;
            T2.weight[(size_t)l] += uw[u];
        }
        for (size_t i = 0; i < added_lcb.size(); i++) added_lcb[i] = ul[(size_t)added_lcb[i]];
        for (int64_t l = 0; l < nl0; l++) cur_of_orig[(size_t)l] = ul[(size_t)cur_of_orig[(size_t)l]];
        for (size_t i = 0; i < ext.size(); i++) if (ul[(size_t)T.n + i] >= 0) { added.push(ext.rec(i)); added_lcb.push_back(ul[(size_t)T.n + i]); }
        T = std::move(T2);
    }
    restore();
    lcb_weight = T.weight;
    *nl_io = T.n;
    if (added.empty()) {
        if (trace) fprintf(stderr, "[trace] lcb extension (device): nothing added, %.3f ms\n", now_ms() - t0);
        return MAUVE_OK;
    }
    // ---- the kept matches into the anchor list (device), LCB ids mapped, weights replaced, recursion gaps recounted ----
    const uint32_t nx = (uint32_t)added.size(), na2 = na + nx;
    std::vector<size_t> idx(nx);
    for (uint32_t i = 0; i < nx; i++) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return added.st(x)[0] < added.st(y)[0]; });
    const size_t b_x = (size_t)nx * (2 + (size_t)N) * 4, b_map = (size_t)nl0 * 4, b_w = (size_t)T.n * 8;
    HIPCHK(c, c->pin_ext.ensure(((b_x + b_map + 7) & ~(size_t)7) + b_w + 64));
    int32_t *hx = c->pin_ext.as<int32_t>(), *hxs = hx + nx, *hxl = hxs + (size_t)nx * N, *hmap = hxl + nx;
    for (uint32_t t = 0; t < nx; t++) {
        const size_t i = idx[t];
        hx[t] = (int32_t)added.len(i); hxl[t] = (int32_t)added_lcb[i];
        for (int g = 0; g < N; g++) hxs[(size_t)t * N + g] = (int32_t)added.st(i)[g];
    }
    for (int64_t l = 0; l < nl0; l++) hmap[l] = (int32_t)cur_of_orig[(size_t)l];
    int64_t *hw = reinterpret_cast<int64_t *>(c->pin_ext.as<char>() + ((b_x + b_map + 7) & ~(size_t)7));
    for (int64_t l = 0; l < T.n; l++) hw[l] = T.weight[(size_t)l];
    HIPCHK(c, c->ext_work.ensure(b_x + b_map + 64 + 64));
    HIPCHK(c, c->ch_anch2.ensure((size_t)na2 * (2 + (size_t)N) * 4 + 64));
    HIPCHK(c, c->ch_lw.ensure(b_w + 64));
    int32_t *dx = c->ext_work.as<int32_t>(), *dxs = dx + nx, *dxl = dxs + (size_t)nx * N, *dmap = dxl + nx;
    uint32_t *dcnt = reinterpret_cast<uint32_t *>(c->ext_work.as<char>() + ((b_x + b_map + 15) & ~(size_t)15));
    int32_t *olen = c->ch_anch2.as<int32_t>(), *ost = olen + na2, *olcb = ost + (size_t)na2 * N;
    HIPCHK(c, hipMemcpyAsync(dx, hx, b_x + b_map, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->ch_lw.p, hw, b_w, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(dcnt, 0, 4, c->stream));
    hipLaunchKernelGGL(ext_merge, dim3((na2 + 255) / 256), dim3(256), 0, c->stream, alen, ast, alcb, na, N, dx, dxs, dxl, nx, dmap, olen, ost, olcb);
    hipLaunchKernelGGL(ext_count_rec, dim3((na2 + 255) / 256), dim3(256), 0, c->stream, olen, ost, olcb, na2, N, p->min_recursive_gap, dcnt);
    HIPCHK(c, hipGetLastError());
    uint32_t *hcnt = reinterpret_cast<uint32_t *>(c->pin_ext.as<char>() + ((b_x + b_map + 7) & ~(size_t)7) + b_w);
    HIPCHK(c, hipMemcpyAsync(hcnt, dcnt, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *alen_io = olen; *ast_io = ost; *alcb_io = olcb; *na_io = na2; *n_rec_io = hcnt[0];
    if (trace) fprintf(stderr, "[trace] lcb extension (device): %u matches added to %u anchors, %lld LCBs, %.3f ms\n", nx, na, (long long)T.n, now_ms() - t0);
    return MAUVE_OK;
}
