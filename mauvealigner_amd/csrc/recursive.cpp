// recursive.cpp -- recursive anchoring of long inter-anchor gaps (Aligner::align's recursion on gaps longer
// than min_recursive_gap_length, mauveAligner.cpp:127,670-672,899; default 200).  Frozen spec: DESIGN.md S8.
//
// MI355X-first shape: instead of one small seed search per gap, every gap of a recursion level that
// wants the same seed weight is searched in ONE batched seed pass.  The gap sub-sequences (in LCB
// orientation) are concatenated per genome into a virtual genome set; the kernels run in segmented mode
// (seed_pass.hip): keys carry the gap id, windows may not straddle gaps, extension stops at gap ends.
// The per-gap chaining that follows (forward-only filter, overlap elimination, collinear LCB) is small
// sequential host work.
#include "common.hpp"
#include <algorithm>
#include <cstdlib>
#include <map>

namespace {

inline uint8_t base_at(const std::vector<uint64_t> &w, int64_t i) { return (uint8_t)((w[(size_t)(i >> 5)] >> (2 * (i & 31))) & 3); }

// A gap waiting for a recursive search: flat record [lcb, prev_w, a(1+N), b(1+N)] in `work`.
struct WorkList {
    int N; std::vector<int64_t> d;
    explicit WorkList(int n) : N(n) {}
    size_t rec() const { return 2 + 2 * (size_t)(1 + N); }
    size_t size() const { return d.size() / rec(); }
    void push(int64_t lcb, int prev_w, const int64_t *a, const int64_t *b)
    {
        d.push_back(lcb); d.push_back(prev_w); d.insert(d.end(), a, a + 1 + N); d.insert(d.end(), b, b + 1 + N);
    }
    int64_t lcb(size_t i) const { return d[i * rec()]; }
    int prev_w(size_t i) const { return (int)d[i * rec() + 1]; }
    const int64_t *a(size_t i) const { return &d[i * rec() + 2]; }
    const int64_t *b(size_t i) const { return &d[i * rec() + 2 + 1 + N]; }
};

inline void gap_of(const int64_t *a, const int64_t *b, int g, int64_t &lo, int64_t &len)
{
    const int64_t sa = a[1 + g], sb = b[1 + g];
    int64_t hi;
    if (sa > 0) { lo = sa + a[0]; hi = sb - 1; }
    else { lo = -sb + b[0]; hi = -sa - 1; }
    len = hi - lo + 1; if (len < 0) len = 0;
}

// which seed weight a gap wants at this level (0 = none), DESIGN.md S8
inline int gap_weight(int N, const int64_t *a, const int64_t *b, int prev_w, int64_t min_gap)
{
    int64_t mx = 0, mn = -1, sum = 0;
    for (int g = 0; g < N; g++) {
        int64_t lo, ln; gap_of(a, b, g, lo, ln);
        mx = std::max(mx, ln); mn = mn < 0 ? ln : std::min(mn, ln); sum += ln;
    }
    if (mx <= min_gap) return 0;
    int w = mauve_default_seed_weight(sum / N);
    if (w > prev_w - 2) w = prev_w - 2;
    if (w < 5) return 0;
    if (mn < mauve_seed_length(mauve_get_seed(w, 0))) return 0;
    return w;
}

}  // namespace

int recursive_anchoring(mauve_ctx *c, const mauve_params *p, int w0, std::vector<MatchVec> &chains, int N, const int *gmap)
{
    const uint32_t full = N >= 32 ? 0xffffffffu : ((1u << N) - 1);
    static const bool trace = getenv("MAUVE_TRACE") != nullptr;
    WorkList work(N);
    for (size_t l = 0; l < chains.size(); l++)
        for (size_t i = 0; i + 1 < chains[l].size(); i++)
            if (gap_weight(N, chains[l].rec(i), chains[l].rec(i + 1), w0, p->min_recursive_gap))
                work.push((int64_t)l, w0, chains[l].rec(i), chains[l].rec(i + 1));
    std::vector<MatchVec> found(chains.size(), MatchVec(N));
    int level = 0;

    while (work.size()) {
        std::map<int, std::vector<size_t>> classes;     // seed weight -> gaps of this level
        for (size_t i = 0; i < work.size(); i++) {
            int w = gap_weight(N, work.a(i), work.b(i), work.prev_w(i), p->min_recursive_gap);
            if (w) classes[w].push_back(i);
        }
        WorkList next(N);
        level++;
        for (auto &cls : classes) {
            const int w = cls.first;
            const std::vector<size_t> &ids = cls.second;
            const uint32_t K = (uint32_t)ids.size();
            const double tc0 = now_ms();
            const uint64_t pat = mauve_get_seed(w, 0);
            // ---- virtual genomes: per genome, the gap sub-sequences in LCB orientation, concatenated ----
            GenomeSet vs; vs.buf = &c->rec_genomes; vs.nseq = N; vs.lens.assign(N, 0); vs.word_off.assign(N, 0);
            std::vector<uint32_t> seg((size_t)N * (K + 1));
            std::vector<int64_t> glo((size_t)N * K), glen((size_t)N * K);
            std::vector<std::vector<uint64_t>> packed(N);
            size_t words = 0;
            for (int g = 0; g < N; g++) {
                int64_t tot = 0;
                for (uint32_t k = 0; k < K; k++) {
                    gap_of(work.a(ids[k]), work.b(ids[k]), g, glo[(size_t)g * K + k], glen[(size_t)g * K + k]);
                    seg[(size_t)g * (K + 1) + k] = (uint32_t)tot; tot += glen[(size_t)g * K + k];
                }
                seg[(size_t)g * (K + 1) + K] = (uint32_t)tot;
                if (tot >= (1LL << 31)) { c->err = "recursive anchoring: gap set too large"; return MAUVE_ERR_LIMIT; }
                vs.lens[g] = tot;
                std::vector<uint8_t> codes((size_t)tot + 1);
                const auto &hw = c->host_packed[gmap ? gmap[g] : g];
                for (uint32_t k = 0; k < K; k++) {
                    uint8_t *out = codes.data() + seg[(size_t)g * (K + 1) + k];
                    const int64_t lo0 = glo[(size_t)g * K + k] - 1, n = glen[(size_t)g * K + k];
                    if (work.a(ids[k])[1 + g] > 0) for (int64_t i = 0; i < n; i++) out[i] = base_at(hw, lo0 + i);
                    else for (int64_t i = 0; i < n; i++) out[i] = (uint8_t)(3 - base_at(hw, lo0 + n - 1 - i));
                }
                packed[g].assign(mauve_packed_words(tot), 0);
                mauve_pack_codes(codes.data(), tot, packed[g].data());
                vs.word_off[g] = words; words += packed[g].size();
            }
            HIPCHK(c, c->rec_genomes.ensure((words + 4) * sizeof(uint64_t)));
            HIPCHK(c, c->rec_seg.ensure(seg.size() * sizeof(uint32_t)));
            for (int g = 0; g < N; g++)
                HIPCHK(c, hipMemcpyAsync(c->rec_genomes.as<uint64_t>() + vs.word_off[g], packed[g].data(),
                                         packed[g].size() * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(c->rec_seg.p, seg.data(), seg.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));    // packed[] must outlive the copies
            int64_t nm = 0;
            int rc = seedpass_run(c, vs, pat, MAUVE_MODE_MEM, full, 1, c->rec_seg.as<uint32_t>(), K, &nm);
            if (rc) return rc;
            if (trace) fprintf(stderr, "[trace] recursion level %d weight %d: %u gaps, %lld bases, %lld matches, %.3f ms\n", level, w, K,
                               (long long)vs.lens[0], (long long)nm, now_ms() - tc0);
            // ---- per-gap chaining of the N-way forward matches ----
            const uint32_t *seg0 = seg.data();
            int64_t i = 0;
            while (i < nm) {
                const int64_t s0 = c->match_start[(size_t)i * N];      // genome 0 is always forward
                const uint32_t k = (uint32_t)(std::upper_bound(seg0, seg0 + K + 1, (uint32_t)(s0 - 1)) - seg0) - 1;
                MatchVec loc(N);
                while (i < nm && (uint32_t)(c->match_start[(size_t)i * N] - 1) < seg0[k + 1]) {
                    bool fwd = true;
                    int64_t rec[1 + MAUVE_MAX_SEQ]; rec[0] = c->match_len[(size_t)i];
                    for (int g = 0; g < N; g++) {
                        int64_t s = c->match_start[(size_t)i * N + g];
                        if (s <= 0) { fwd = false; break; }
                        rec[1 + g] = s - seg[(size_t)g * (K + 1) + k];     // 1-based inside the gap
                    }
                    if (fwd) loc.push(rec);
                    i++;
                }
                if (loc.empty()) continue;
                ChainOrders orders;
                host_eliminate_overlaps(loc, &orders);
                std::vector<int64_t> ml; int64_t nl = 0;
                host_lcb_chain(loc, 0, true, ml, nl, &orders);
                const size_t wi = ids[k];
                const int64_t *A = work.a(wi);
                MatchVec glob(N);
                for (size_t q = 0; q < loc.size(); q++) {
                    if (ml[q] < 0) continue;
                    int64_t rec[1 + MAUVE_MAX_SEQ]; rec[0] = loc.len(q);
                    for (int g = 0; g < N; g++) {
                        const int64_t s = loc.st(q)[g], lo = glo[(size_t)g * K + k], ln = glen[(size_t)g * K + k];
                        if (A[1 + g] > 0) rec[1 + g] = lo + s - 1;
                        else { const int64_t hi = lo + ln - 1; rec[1 + g] = -(hi - (s - 1) - rec[0] + 1); }
                    }
                    glob.push(rec);
                }
                if (glob.empty()) continue;
                glob.sort_by_start0();
                const int64_t lcb = work.lcb(wi);
                for (size_t q = 0; q <= glob.size(); q++)
                    next.push(lcb, w, q == 0 ? work.a(wi) : glob.rec(q - 1), q == glob.size() ? work.b(wi) : glob.rec(q));
                for (size_t q = 0; q < glob.size(); q++) found[(size_t)lcb].push(glob.rec(q));
            }
        }
        work.d.swap(next.d);
    }
    for (size_t l = 0; l < chains.size(); l++) {
        if (found[l].empty()) continue;
        chains[l].d.insert(chains[l].d.end(), found[l].d.begin(), found[l].d.end());
        chains[l].sort_by_start0();
    }
    return MAUVE_OK;
}
