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

__global__ void __launch_bounds__(64) bb_pair_gaps(const uint32_t *__restrict__ cols, const BbIv *__restrict__ ivs, uint32_t n_ivs, uint32_t island_gap,
                                                    BbRec *__restrict__ rec, uint32_t cap, uint32_t *__restrict__ count)
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
        for (int64_t w = cs - 64; w > -64; w -= 64) {
            const int64_t c = w + lane;
            const uint32_t v = c >= 0 ? m[c] : 0u;
            const bool ra = v >> a & 1, rb = v >> b & 1;
            const uint64_t any = __ballot(ra || rb), both = __ballot(ra && rb);
            if (any) { const int top = 63 - __clzll((long long)any); if (both >> top & 1) seen_both = true; else skipping = true; break; }
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
    const size_t o_cnt = up(ivs.size() * sizeof(BbIv)), o_tile = o_cnt + 64, o_rec = o_tile + up(n_tiles * (size_t)N * 4);
    size_t cap = std::max<size_t>(1u << 16, c->bb_rec_cap);
    std::vector<BbRec> recs;
    std::vector<uint32_t> tile_cnt(n_tiles * (size_t)N);
    for (int attempt = 0;; attempt++) {
        HIPCHK(c, c->bb_work.ensure(o_rec + cap * sizeof(BbRec)));
        char *wk = c->bb_work.as<char>();
        HIPCHK(c, hipMemcpyAsync(wk, ivs.data(), ivs.size() * sizeof(BbIv), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemsetAsync(wk + o_cnt, 0, 64, c->stream));
        const uint32_t gy = std::min<uint32_t>(max_pairs, 64);        // one wave per (chunk, pair): with few pairs, too, every resident wave works
        hipLaunchKernelGGL(bb_pair_gaps, dim3(chunks, gy), dim3(64), 0, c->stream, d_cols, reinterpret_cast<const BbIv *>(wk), (uint32_t)ivs.size(),
                           (uint32_t)island_gap, reinterpret_cast<BbRec *>(wk + o_rec), (uint32_t)cap, reinterpret_cast<uint32_t *>(wk + o_cnt));
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
    const int rc = backbone_run(c, nseq, n_iv, left, right, reverse, col_off, c->bb_cols.as<uint32_t>(), island_gap_size);
    if (rc) return rc;
    *n_seg = (int64_t)c->bb.seg_iv.size(); *n_islands = (int64_t)c->bb.islands.size() / 8;
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
