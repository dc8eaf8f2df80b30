// assemble_dev.hip -- the interval table of mauve_align assembled on the device.
//
// GappedMatchRecord::finalize's last step (MatchRecord.h:323-345: the anchors and the aligned stretches between them are
// spliced into one alignment per LCB) and addUnalignedIntervals (mauveAligner.cpp:748), frozen layout DESIGN.md S6:
// per LCB the columns of anchor 0, the stretch behind it, anchor 1, ... ; then the single-genome islands.
//
// When the chains never left the device (chain_order_device) and the DP results stayed there (dp_run_from_anchors with
// stay_on_device), nothing of the result has to cross PCIe for the pass to finish: the column array (the bulk: one
// uint32 per alignment column) is written here by a wave per anchor, the per-LCB rows of the interval table (column
// offsets, extents, DP score sums) come back in one small copy, and the columns and the anchor table are copied to the
// host only when the caller fetches them (materialize_result).
#include "common.hpp"
#include "dev_scan.hpp"
#include <cstring>
#include <cstdlib>
#include <algorithm>

namespace {
using namespace devscan;

// the stretch between anchor k and k + 1 of one chain in genome g (same arithmetic as gap_of, pipeline.cpp / dpf_gap)
__device__ __forceinline__ int64_t as_gap_len(const int32_t *__restrict__ alen, const int32_t *__restrict__ ast, int N, uint32_t k, int g)
{
    const int64_t sa = ast[(size_t)k * N + g], sb = ast[(size_t)(k + 1) * N + g];
    int64_t lo, hi;
    if (sa > 0) { lo = sa + alen[k]; hi = sb - 1; }
    else { lo = -sb + alen[k + 1]; hi = -sa - 1; }
    const int64_t ln = hi - lo + 1;
    return ln < 0 ? 0 : ln;
}

// columns of anchor a and of the stretch behind it
struct AsWidth {
    const int32_t *alen, *ast, *gapcode; const int64_t *dcol_off; int N;
    __device__ int64_t value(uint32_t a) const
    {
        int64_t w = alen[a];
        const int32_t code = gapcode[a];
        if (code >= 0) w += dcol_off[code + 1] - dcol_off[code];
        else if (code == -2) for (int g = 0; g < N; g++) w += as_gap_len(alen, ast, N, a, g);
        return w;
    }
};

// first / last anchor of every LCB, the LCB's first column, its DP score sum
__global__ void __launch_bounds__(256) as_lcb(const int32_t *__restrict__ alcb, const int32_t *__restrict__ gapcode, const int64_t *__restrict__ dscore,
                                              const int64_t *__restrict__ col0, uint32_t na, uint32_t *__restrict__ first_a, uint32_t *__restrict__ last_a,
                                              int64_t *__restrict__ lcb_col, unsigned long long *__restrict__ lcb_score)
{
    const uint32_t a = blockIdx.x * 256u + threadIdx.x;
    const bool in = a < na;
    const int32_t l = in ? alcb[a] : 0;
    const int32_t code = in ? gapcode[a] : -1;
    wave_keyed_add(lcb_score, code >= 0, (uint32_t)l, code >= 0 ? (unsigned long long)dscore[code] : 0ull);
    if (!in) return;
    if (a == 0 || alcb[a - 1] != l) { first_a[l] = a; lcb_col[l] = col0[a]; }
    if (a + 1 == na || alcb[a + 1] != l) last_a[l] = a;
}

// LCB extents: the anchors of a chain are ordered, so the ends come from its first and last anchor (signed: negative = reverse)
// (also: the LCB weights and the total number of LCB columns into the block of rows that goes to the host in one copy)
__global__ void __launch_bounds__(256) as_extents(const int32_t *__restrict__ alen, const int32_t *__restrict__ ast, int N, uint32_t nl,
                                                  const uint32_t *__restrict__ first_a, const uint32_t *__restrict__ last_a,
                                                  int64_t *__restrict__ left, int64_t *__restrict__ right,
                                                  const unsigned long long *__restrict__ lw, int64_t *__restrict__ lw_row,
                                                  const int64_t *__restrict__ total_cols, int64_t *__restrict__ total_row)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= nl * (uint32_t)N) return;
    const uint32_t l = t / (uint32_t)N; const int g = (int)(t % (uint32_t)N);
    if (g == 0) lw_row[l] = lw ? (int64_t)lw[l] : 0;
    if (t == 0) *total_row = *total_cols;
    const uint32_t a0 = first_a[l], a1 = last_a[l];
    const int64_t s0 = ast[(size_t)a0 * N + g], s1 = ast[(size_t)a1 * N + g];
    int64_t le, re;
    if (s0 > 0) { le = s0; re = s1 + alen[a1] - 1; }
    else { le = -s1; re = -s0 + alen[a0] - 1; }
    left[t] = s0 < 0 ? -le : le; right[t] = s0 < 0 ? -re : re;
}

// one wave per anchor: its own columns (every genome present), then the stretch behind it -- the DP's columns, or, for a
// stretch that was not aligned, its bases genome by genome
__global__ void __launch_bounds__(256) as_fill(const int32_t *__restrict__ alen, const int32_t *__restrict__ ast, const int32_t *__restrict__ gapcode,
                                               const int64_t *__restrict__ dcol_off, const uint32_t *__restrict__ dcols,
                                               const int64_t *__restrict__ col0, uint32_t na, int N, uint32_t full, uint32_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nw = (gridDim.x * 256u) >> 6;
    for (uint32_t a = wave; a < na; a += nw) {
        uint32_t *o = out + col0[a];
        const int32_t ln = alen[a];
        for (int32_t c = lane; c < ln; c += 64) o[c] = full;
        o += ln;
        const int32_t code = gapcode[a];
        if (code >= 0) {
            const int64_t s0 = dcol_off[code], n = dcol_off[code + 1] - s0;
            for (int64_t c = lane; c < n; c += 64) o[c] = dcols[s0 + c];
        } else if (code == -2) {
            for (int g = 0; g < N; g++) {
                const int64_t n = as_gap_len(alen, ast, N, a, g);
                for (int64_t c = lane; c < n; c += 64) o[c] = 1u << g;
                o += n;
            }
        }
    }
}

struct Island { int64_t col, len; uint32_t bit, pad; };
__global__ void __launch_bounds__(256) as_islands(const Island *__restrict__ isl, uint32_t n, uint32_t *__restrict__ out)
{
    for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
        const Island is = isl[k];
        uint32_t *o = out + is.col;
        for (int64_t c = threadIdx.x; c < is.len; c += 256) o[c] = is.bit;
    }
}

// ---- extant sum-of-pairs score of ungapped matches ---------------------------------------------------------
// One wave per match: a lane takes every 64th column, reads the base of every present component from the packed genomes
// (a reverse component from its right end, complemented), adds the substitution scores of all pairs; wave reduction.
struct SpGenomes { uint64_t word_off[MAUVE_MAX_SEQ]; int32_t s[4][4]; };
__global__ void __launch_bounds__(256) sp_score_matches(const uint64_t *__restrict__ packed, SpGenomes G, int n, const int64_t *__restrict__ rec,
                                                        uint32_t nm, int64_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nw = (gridDim.x * 256u) >> 6;
    for (uint32_t i = wave; i < nm; i += nw) {
        const int64_t *r = rec + (size_t)i * (1 + n);
        const int64_t len = r[0];
        int64_t acc = 0;
        for (int64_t c = lane; c < len; c += 64) {
            uint32_t have = 0, bases = 0;                      // 2 bits per component
            for (int g = 0; g < n; g++) {
                const int64_t st = r[1 + g];
                if (!st) continue;
                const int64_t p = st > 0 ? st - 1 + c : -st - 1 + (len - 1 - c);
                uint32_t b = (uint32_t)(packed[G.word_off[g] + (uint64_t)(p >> 5)] >> (2 * (p & 31))) & 3u;
                if (st < 0) b = 3u - b;
                have |= 1u << g; bases |= b << (2 * g);          // n <= 16 (checked on the host)
            }
            for (int x = 0; x < n; x++) {
                if (!(have >> x & 1)) continue;
                const uint32_t bx = (bases >> (2 * x)) & 3u;
                for (int y = x + 1; y < n; y++) if (have >> y & 1) acc += G.s[bx][(bases >> (2 * y)) & 3u];
            }
        }
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) out[i] = acc;
    }
}

// the anchor table in the caller's width (int64), for a fetch straight into page-locked caller buffers
__global__ void __launch_bounds__(256) as_widen_anchors(const int32_t *__restrict__ alen, const int32_t *__restrict__ ast, const int32_t *__restrict__ alcb, uint32_t na, int N,
                                                        int64_t *__restrict__ olen, int64_t *__restrict__ ost, int64_t *__restrict__ olcb)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < na) { olen[i] = alen[i]; olcb[i] = alcb[i]; }
    if (i < na * (uint32_t)N) ost[i] = ast[i];
}

}  // namespace

int64_t sp_default_min_weight(int w, int n, const mauve_scoring *sc)
{
    int32_t d = sc->matrix[0][0];
    for (int a = 1; a < 4; a++) d = std::min(d, sc->matrix[a][a]);
    return (int64_t)3 * w * ((int64_t)n * (n - 1) / 2) * d;
}

int match_sp_scores(mauve_ctx *c, const MatchVec &m, const int *gmap, const mauve_scoring *sc, std::vector<int64_t> &out)
{
    const int n = m.N; const size_t nm = m.size();
    out.assign(nm, 0);
    if (nm == 0) return MAUVE_OK;
    if (n > 16) { c->err = "sum-of-pairs LCB scoring: at most 16 genomes"; return MAUVE_ERR_LIMIT; }
    for (size_t i = 0; i < nm; i++)
        for (int g = 0; g < n; g++) {
            const int64_t st = m.st(i)[g], L = c->lens[(size_t)(gmap ? gmap[g] : g)];
            if (st && m.len(i) > 0 && std::llabs(st) + m.len(i) - 1 > L) { c->err = "sp scores: match outside its genome"; return MAUVE_ERR_ARG; }   // (records of length <= 0 score 0)
        }
    SpGenomes G; memset(&G, 0, sizeof G);
    for (int g = 0; g < n; g++) G.word_off[g] = c->word_off[(size_t)(gmap ? gmap[g] : g)];
    memcpy(G.s, sc->matrix, sizeof G.s);
    const size_t rb = nm * (1 + (size_t)n) * 8;
    HIPCHK(c, c->as_work.ensure(rb + nm * 8 + 64));
    HIPCHK(c, c->pin_asm.ensure(rb + nm * 8 + 64));
    int64_t *d_rec = c->as_work.as<int64_t>(), *d_out = d_rec + nm * (1 + (size_t)n);
    memcpy(c->pin_asm.p, m.d.data(), rb);
    HIPCHK(c, hipMemcpyAsync(d_rec, c->pin_asm.p, rb, hipMemcpyHostToDevice, c->stream));
    const uint32_t blocks = (uint32_t)std::min<size_t>((nm + 3) / 4, 256 * 8);
    hipLaunchKernelGGL(sp_score_matches, dim3(blocks), dim3(256), 0, c->stream, c->genomes.as<uint64_t>(), G, n, d_rec, (uint32_t)nm, d_out);
    HIPCHK(c, hipGetLastError());
    int64_t *h_out = reinterpret_cast<int64_t *>(c->pin_asm.as<char>() + rb);
    HIPCHK(c, hipMemcpyAsync(h_out, d_out, nm * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    memcpy(out.data(), h_out, nm * 8);
    return MAUVE_OK;
}

// Everything of align_finish, from the device-side chains and DP results.  The host receives the per-LCB rows; the
// columns stay in c->res_cols, the anchor table where the chain stage left it, until materialize_result.
int assemble_device(mauve_ctx *c, int64_t na64, int64_t cells, mauve_align_sizes *sizes, bool host_chains)
{
    static const bool trace = getenv("MAUVE_TRACE") != nullptr;
    AlignState &S = c->ast; AlignResult &R = c->res;
    const int N = S.N; const int64_t nl = S.nl; const uint32_t na = (uint32_t)na64;
    const mauve_ctx::DpFrontOut &fo = c->dpf_out;
    const double t0 = now_ms();
    c->stage.dp_ms = t0 - S.t_dp0;
    const uint32_t nb = (na + TILE - 1) / TILE, blocks = (na + 255) / 256;
    // work area: col0[na + 1], tile sums, per LCB: first / last anchor, first column, extents, score
    auto up8 = [](size_t x) { return (x + 7) & ~(size_t)7; };
    const size_t o_bsum = ((size_t)na + 1) * 8, o_first = o_bsum + ((size_t)nb + 8) * 8, o_last = up8(o_first + (size_t)nl * 4),
                 o_col = up8(o_last + (size_t)nl * 4), o_left = o_col + ((size_t)nl + 1) * 8, o_right = o_left + (size_t)nl * N * 8,
                 o_score = o_right + (size_t)nl * N * 8, o_lw = o_score + (size_t)nl * 8, o_tot = o_lw + (size_t)nl * 8, w_total = o_tot + 8;
    HIPCHK(c, c->as_work.ensure(w_total + 64));
    HIPCHK(c, c->res_cols.ensure(((size_t)S.sum + 64) * 4));          // every column holds a base, every base sits in one column
    char *wk = c->as_work.as<char>();
    int64_t *col0 = reinterpret_cast<int64_t *>(wk), *bsum = reinterpret_cast<int64_t *>(wk + o_bsum);
    uint32_t *first_a = reinterpret_cast<uint32_t *>(wk + o_first), *last_a = reinterpret_cast<uint32_t *>(wk + o_last);
    int64_t *lcb_col = reinterpret_cast<int64_t *>(wk + o_col), *left = reinterpret_cast<int64_t *>(wk + o_left),
            *right = reinterpret_cast<int64_t *>(wk + o_right);
    unsigned long long *score = reinterpret_cast<unsigned long long *>(wk + o_score);
    int64_t *lw_row = reinterpret_cast<int64_t *>(wk + o_lw), *tot_row = reinterpret_cast<int64_t *>(wk + o_tot);
    uint32_t *out = c->res_cols.as<uint32_t>();
    HIPCHK(c, hipMemsetAsync(score, 0, (size_t)nl * 8, c->stream));
    // the DP front end may have found no interval: its offset / score arrays are then not set up
    const AsWidth wf{fo.alen, fo.ast, fo.gapcode, fo.col_off, N};
    hipLaunchKernelGGL((vscan_partial<int64_t, AsWidth>), dim3(nb), dim3(256), 0, c->stream, wf, na, bsum);
    hipLaunchKernelGGL((vscan_write<int64_t, AsWidth>), dim3(nb), dim3(256), 0, c->stream, wf, na, bsum, col0, (int64_t *)nullptr);
    hipLaunchKernelGGL(as_lcb, dim3(blocks), dim3(256), 0, c->stream, fo.alcb, fo.gapcode, fo.score, col0, na, first_a, last_a, lcb_col, score);
    hipLaunchKernelGGL(as_extents, dim3(((uint32_t)nl * N + 255) / 256), dim3(256), 0, c->stream, fo.alen, fo.ast, N, (uint32_t)nl, first_a, last_a, left, right,
                       host_chains ? (const unsigned long long *)nullptr : c->ch_lw.as<unsigned long long>(), lw_row, col0 + na, tot_row);
    HIPCHK(c, hipGetLastError());
    // per-LCB rows to the host: first columns, extents, scores, weights, and the number of LCB columns
    const size_t rows_bytes = ((size_t)nl + 1) * 8 + 2 * (size_t)nl * N * 8 + 2 * (size_t)nl * 8 + 8;
    HIPCHK(c, c->pin_asm.ensure(64 + rows_bytes));
    int64_t *h_rows = c->pin_asm.as<int64_t>() + 8;
    HIPCHK(c, hipMemcpyAsync(h_rows, lcb_col, rows_bytes, hipMemcpyDeviceToHost, c->stream));       // lcb_col .. total are adjacent
    {
        const uint32_t fb = (uint32_t)std::min<int64_t>(((int64_t)na + 3) / 4, 256 * 8);
        hipLaunchKernelGGL(as_fill, dim3(fb), dim3(256), 0, c->stream, fo.alen, fo.ast, fo.gapcode, fo.col_off, fo.cols, col0, na, N, S.full, out);
    }
    // The match list (if the seed pass left it on the device only: sorted_rec) and the anchor table (ch_anch) are
    // fetched from where they are: both stay untouched until this context's next seed pass / chain.
    // (host_chains: both are on the host already -- the chains came from there)
    R.dev_nm = !host_chains && c->matches_pending ? (size_t)c->n_matches : 0;
    if (S.mums_kept >= 0) {                  // the main pass's list was set aside for the recursion (pipeline.cpp): back in place, fetched from there
        std::swap(c->sorted_rec, c->sorted_rec_keep);
        R.dev_nm = (size_t)S.mums_kept; S.mums_kept = -1;
    }
    R.dev_alen = fo.alen; R.dev_ast = fo.ast; R.dev_alcb = fo.alcb;
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const double t1 = now_ms();
    const int64_t *h_col = h_rows, *h_left = h_col + nl + 1, *h_right = h_left + nl * N, *h_score = h_right + nl * N, *h_lw = h_score + nl;
    const int64_t lcb_cols = h_lw[nl];
    R.col_off.assign(h_col, h_col + nl);
    R.lcb_left.assign(h_left, h_left + nl * N); R.lcb_right.assign(h_right, h_right + nl * N);
    R.dp_score.assign(h_score, h_score + nl);
    if (!host_chains && !S.lw_from_host) R.lcb_weight.assign(h_lw, h_lw + nl);      // (host chains, device extension: filled by align_begin)
    R.iv_left.assign((size_t)nl * N, 0); R.iv_right.assign((size_t)nl * N, 0); R.iv_reverse.assign((size_t)nl * N, 0);
    for (int64_t i = 0; i < nl * N; i++) {
        R.iv_left[(size_t)i] = std::llabs(R.lcb_left[(size_t)i]);
        R.iv_right[(size_t)i] = std::llabs(R.lcb_right[(size_t)i]);
        R.iv_reverse[(size_t)i] = R.lcb_left[(size_t)i] < 0;
    }
    int64_t ncols = lcb_cols, niv = nl;
    if (S.p.add_unaligned) {
        // islands: same sweep as align_finish; the columns are filled on the device from a small table
        std::vector<Island> isl;
        for (int g = 0; g < N; g++) {
            std::vector<std::pair<int64_t, int64_t>> sp;
            for (int64_t l = 0; l < nl; l++) if (R.iv_left[(size_t)l * N + g]) sp.push_back({R.iv_left[(size_t)l * N + g], R.iv_right[(size_t)l * N + g]});
            std::sort(sp.begin(), sp.end());
            int64_t cur = 1;
            for (size_t i = 0; i <= sp.size(); i++) {
                const int64_t lo = cur, hi = i < sp.size() ? sp[i].first - 1 : c->lens[(size_t)g];
                if (hi >= lo) {
                    R.col_off.push_back(ncols);
                    Island is; is.col = ncols; is.len = hi - lo + 1; is.bit = 1u << g; is.pad = 0;
                    isl.push_back(is);
                    ncols += hi - lo + 1;
                    for (int h = 0; h < N; h++) { R.iv_left.push_back(h == g ? lo : 0); R.iv_right.push_back(h == g ? hi : 0); R.iv_reverse.push_back(0); }
                    R.dp_score.push_back(0);
                    niv++;
                }
                if (i < sp.size() && sp[i].second + 1 > cur) cur = sp[i].second + 1;
            }
        }
        if (ncols > S.sum) { c->err = "assemble_device: more columns than bases (internal error)"; return MAUVE_ERR_HIP; }
        if (!isl.empty()) {
            const size_t ib = isl.size() * sizeof(Island);
            HIPCHK(c, c->pin_asm.ensure(64 + rows_bytes + ib + 64));
            char *pi = c->pin_asm.as<char>() + 64 + rows_bytes;
            memcpy(pi, isl.data(), ib);
            HIPCHK(c, c->as_isl.ensure(ib + 64));
            HIPCHK(c, hipMemcpyAsync(c->as_isl.p, pi, ib, hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(as_islands, dim3((uint32_t)std::min<size_t>(isl.size(), 1024)), dim3(256), 0, c->stream,
                               c->as_isl.as<Island>(), (uint32_t)isl.size(), out);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipStreamSynchronize(c->stream));            // pin_asm is reused by the next call
        }
    }
    R.col_off.push_back(ncols);
    R.n_cols = (size_t)ncols;
    if (host_chains && !S.anchor_table_done) {                       // anchor_length / _start / _lcb straight from the host chains
        R.anchor_length.resize((size_t)na); R.anchor_start.resize((size_t)na * N); R.anchor_lcb.resize((size_t)na);
        size_t a = 0;
        for (int64_t l = 0; l < nl; l++) {
            const MatchVec &ch = S.chains[(size_t)l];
            for (size_t i = 0; i < ch.size(); i++, a++) {
                R.anchor_length[a] = ch.len(i); R.anchor_lcb[a] = l;
                std::copy(ch.st(i), ch.st(i) + N, &R.anchor_start[a * N]);
            }
        }
    }
    R.dev_pending = true; R.cols_pending = true; R.dev_na = host_chains ? 0 : na; R.cols_ext = nullptr;
    R.cols_fill = 0; R.cols_dirty.clear();                       // the host column buffer no longer holds the "all anchors" state
    R.sz.n_mums = S.nm; R.sz.n_lcb = nl; R.sz.n_anchor = na; R.sz.n_iv = niv; R.sz.n_cols = ncols;
    R.sz.n_gap_dp = fo.n_dp; R.sz.n_dp_cells = cells;
    *sizes = R.sz;
    const double t2 = now_ms();
    if (trace) fprintf(stderr, "[trace] assemble (device): layout+fill %.3f ms, islands+tables %.3f\n", t1 - t0, t2 - t1);
    c->stage.assemble_ms = t2 - t0;
    c->stage.total_ms = t2 - S.t0;
    S.open = false;
    return MAUVE_OK;
}

// The anchor table and (if the seed pass left it there) the match list of a device-assembled result -> host.  These two
// live in buffers the next seed pass / chain / DP front end reuses (sorted_rec, ch_anch), so every entry point that runs
// such work calls this first; the columns have a buffer of their own (res_cols) that only the next assembly writes.
// Idempotent.
int materialize_tables(mauve_ctx *c)
{
    AlignResult &R = c->res;
    if (!R.dev_pending) return MAUVE_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const int N = c->ast.N; const size_t na = R.dev_na;
    const size_t ab = (na * (2 + (size_t)N) * 4 + 63) & ~(size_t)63, mb = R.dev_nm * (1 + (size_t)N) * 8;
    HIPCHK(c, c->pin_tab.ensure(ab + mb + 64));
    char *pa = c->pin_tab.as<char>();
    if (ab) {
        HIPCHK(c, hipMemcpyAsync(pa, R.dev_alen, na * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(pa + na * 4, R.dev_ast, na * N * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(pa + na * 4 * (1 + (size_t)N), R.dev_alcb, na * 4, hipMemcpyDeviceToHost, c->stream));
    }
    if (mb) HIPCHK(c, hipMemcpyAsync(pa + ab, c->sorted_rec.p, mb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (mb) {
        const int64_t *hm = reinterpret_cast<const int64_t *>(pa + ab);
        R.mum_length.assign(hm, hm + R.dev_nm); R.mum_start.assign(hm + R.dev_nm, hm + R.dev_nm * (1 + (size_t)N));
        R.dev_nm = 0;
    }
    const int32_t *hl = reinterpret_cast<const int32_t *>(pa), *hs = hl + na, *hb = hs + na * N;
    if (na) {                                                    // (0: the anchor table came from host chains and is filled)
        R.anchor_length.resize(na); R.anchor_start.resize(na * N); R.anchor_lcb.resize(na);
        for (size_t a = 0; a < na; a++) { R.anchor_length[a] = hl[a]; R.anchor_lcb[a] = hb[a]; }
        for (size_t i = 0; i < na * N; i++) R.anchor_start[i] = hs[i];
    }
    R.dev_pending = false;
    return MAUVE_OK;
}

// The bulk tables of a device-assembled result straight into the caller's buffers when those are page-locked (mauve_host_alloc):
// the match list is on the device in the caller's layout already (sorted_rec: int64 length[n], start[n * N]), the anchor table
// is widened to int64 there; every copy is one DMA, no staging block, no conversion loop on the host.  Returns true when it
// took care of both tables (the context keeps no host copy of them then: they are still pending for a later call).
bool fetch_tables_direct(mauve_ctx *c, int64_t *mum_length, int64_t *mum_start, int64_t *anchor_length, int64_t *anchor_start, int64_t *anchor_lcb, int *rc_out)
{
    AlignResult &R = c->res;
    *rc_out = MAUVE_OK;
    if (!R.dev_pending) return false;
    const int N = c->ast.N; const size_t na = R.dev_na, nm = R.dev_nm;
    if (nm && !(host_pointer_is_pinned(mum_length) && host_pointer_is_pinned(mum_start))) return false;
    if (na && !(host_pointer_is_pinned(anchor_length) && host_pointer_is_pinned(anchor_start) && host_pointer_is_pinned(anchor_lcb))) return false;
    auto chk = [&](hipError_t e) { if (e != hipSuccess && *rc_out == MAUVE_OK) { c->err = std::string("fetch: ") + hipGetErrorString(e); *rc_out = MAUVE_ERR_HIP; } };
    chk(hipSetDevice(c->device));
    if (na) {
        chk(c->as_wide.ensure(na * (2 + (size_t)N) * 8 + 64));
        if (*rc_out) return true;
        int64_t *wl = c->as_wide.as<int64_t>(), *ws = wl + na, *wb = ws + na * N;
        hipLaunchKernelGGL(as_widen_anchors, dim3((uint32_t)((na * N + 255) / 256)), dim3(256), 0, c->stream, R.dev_alen, R.dev_ast, R.dev_alcb, (uint32_t)na, N, wl, ws, wb);
        chk(hipGetLastError());
        chk(hipMemcpyAsync(anchor_length, wl, na * 8, hipMemcpyDeviceToHost, c->stream));
        chk(hipMemcpyAsync(anchor_start, ws, na * N * 8, hipMemcpyDeviceToHost, c->stream));
        chk(hipMemcpyAsync(anchor_lcb, wb, na * 8, hipMemcpyDeviceToHost, c->stream));
    }
    if (nm) {
        const int64_t *rec = c->sorted_rec.as<int64_t>();
        chk(hipMemcpyAsync(mum_length, rec, nm * 8, hipMemcpyDeviceToHost, c->stream));
        chk(hipMemcpyAsync(mum_start, rec + nm, nm * N * 8, hipMemcpyDeviceToHost, c->stream));
    }
    return true;                                              // (the stream is drained by the column copy that follows, or by the caller)
}

// ---- compact fetch (mauve_align_fetch_compact): the result in the narrowest types that hold it ----
template <typename T>
__global__ void __launch_bounds__(256) as_narrow_cols(const uint32_t *__restrict__ in, size_t n, T *__restrict__ out)
{
    // sixteen columns per thread: four 16-byte loads, one (u8) or two (u16) 16-byte stores
    const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (i0 + 16 <= n) {
        uint4 v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = reinterpret_cast<const uint4 *>(in + i0)[q];
        T o[16];
#pragma unroll
        for (int q = 0; q < 4; q++) { o[4 * q] = (T)v[q].x; o[4 * q + 1] = (T)v[q].y; o[4 * q + 2] = (T)v[q].z; o[4 * q + 3] = (T)v[q].w; }
        uint4 *dst = reinterpret_cast<uint4 *>(out + i0);
#pragma unroll
        for (int q = 0; q < (int)(sizeof(T) * 16 / 16); q++) dst[q] = reinterpret_cast<const uint4 *>(o)[q];
    } else
        for (size_t i = i0; i < n; i++) out[i] = (T)in[i];
}
__global__ void __launch_bounds__(256) as_narrow_i64(const int64_t *__restrict__ in, size_t n, int32_t *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (int32_t)in[i];
}

// The device-resident bulk of a result straight into page-locked caller buffers in compact form: the anchor table as the int32 arrays the chain
// stage left (no widening), the match list narrowed to int32, the columns narrowed to col_bytes each.  Returns false when something is
// not where this needs it (not pending on the device, a pageable buffer): the caller then converts the host copy.
bool fetch_compact_direct(mauve_ctx *c, int col_bytes, int32_t *mum_length, int32_t *mum_start, int32_t *anchor_length, int32_t *anchor_start, int32_t *anchor_lcb,
                          void *cols, bool *tables_done, bool *cols_done, int *rc_out)
{
    AlignResult &R = c->res;
    *rc_out = MAUVE_OK; *tables_done = false; *cols_done = false;
    auto chk = [&](hipError_t e) { if (e != hipSuccess && *rc_out == MAUVE_OK) { c->err = std::string("fetch_compact: ") + hipGetErrorString(e); *rc_out = MAUVE_ERR_HIP; } };
    const int N = c->ast.N; const size_t na = R.dev_na, nm = R.dev_nm;
    bool any = false;
    if (R.dev_pending && mum_length && mum_start && anchor_length && anchor_start && anchor_lcb &&
        (!nm || (host_pointer_is_pinned(mum_length) && host_pointer_is_pinned(mum_start))) &&
        (!na || (host_pointer_is_pinned(anchor_length) && host_pointer_is_pinned(anchor_start) && host_pointer_is_pinned(anchor_lcb)))) {
        chk(hipSetDevice(c->device));
        if (na) {
            chk(hipMemcpyAsync(anchor_length, R.dev_alen, na * 4, hipMemcpyDeviceToHost, c->stream));
            chk(hipMemcpyAsync(anchor_start, R.dev_ast, na * N * 4, hipMemcpyDeviceToHost, c->stream));
            chk(hipMemcpyAsync(anchor_lcb, R.dev_alcb, na * 4, hipMemcpyDeviceToHost, c->stream));
        }
        if (nm) {
            const size_t n = nm * (1 + (size_t)N);
            chk(c->as_wide.ensure(n * 4 + 64));
            if (*rc_out) return true;
            hipLaunchKernelGGL(as_narrow_i64, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, c->sorted_rec.as<int64_t>(), n, c->as_wide.as<int32_t>());
            chk(hipGetLastError());
            chk(hipMemcpyAsync(mum_length, c->as_wide.as<int32_t>(), nm * 4, hipMemcpyDeviceToHost, c->stream));
            chk(hipMemcpyAsync(mum_start, c->as_wide.as<int32_t>() + nm, nm * N * 4, hipMemcpyDeviceToHost, c->stream));
        }
        *tables_done = true; any = true;
    }
    if (cols && R.n_cols && R.cols_pending && host_pointer_is_pinned(cols)) {
        chk(hipSetDevice(c->device));
        const size_t n = R.n_cols;
        if (col_bytes == 4) chk(hipMemcpyAsync(cols, c->res_cols.p, n * 4, hipMemcpyDeviceToHost, c->stream));
        else {
            chk(c->res_narrow.ensure(n * (size_t)col_bytes + 64));
            if (*rc_out) return true;
            const uint32_t blocks = (uint32_t)((n + 4095) / 4096);
            if (col_bytes == 1) hipLaunchKernelGGL(as_narrow_cols<uint8_t>, dim3(blocks), dim3(256), 0, c->stream, c->res_cols.as<uint32_t>(), n, c->res_narrow.as<uint8_t>());
            else hipLaunchKernelGGL(as_narrow_cols<uint16_t>, dim3(blocks), dim3(256), 0, c->stream, c->res_cols.as<uint32_t>(), n, c->res_narrow.as<uint16_t>());
            chk(hipGetLastError());
            chk(hipMemcpyAsync(cols, c->res_narrow.p, n * (size_t)col_bytes, hipMemcpyDeviceToHost, c->stream));
        }
        *cols_done = true; any = true;
    }
    if (any) chk(hipStreamSynchronize(c->stream));
    return any;
}

// ... and the columns into page-locked staging (the XMFA writer, a fetch into pageable memory); idempotent
int materialize_result(mauve_ctx *c)
{
    int rc = materialize_tables(c);
    if (rc) return rc;
    AlignResult &R = c->res;
    if (!R.cols_pending) return MAUVE_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, c->pin_cols.ensure(R.n_cols * 4 + 64));
    if (R.n_cols) {
        HIPCHK(c, hipMemcpyAsync(c->pin_cols.p, c->res_cols.p, R.n_cols * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    R.cols_ext = c->pin_cols.as<uint32_t>();
    R.cols_pending = false;
    return MAUVE_OK;
}

// the columns of the result into the caller's buffer: one DMA straight from HBM when the buffer is page-locked
// (mauve_host_alloc) and the columns are still there; otherwise through the staging copy
int fetch_columns(mauve_ctx *c, uint32_t *dst)
{
    AlignResult &R = c->res;
    if (!R.n_cols) return MAUVE_OK;
    if (R.cols_pending && host_pointer_is_pinned(dst)) {
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipMemcpyAsync(dst, c->res_cols.p, R.n_cols * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return MAUVE_OK;
    }
    int rc = materialize_result(c);
    if (rc) return rc;
    memcpy(dst, R.cols_data(), R.n_cols * sizeof(uint32_t));
    return MAUVE_OK;
}
