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

struct RGap {
    int64_t lcb;
    HMatch a, b;
    int prev_w;
    int64_t lo[MAUVE_MAX_SEQ], len[MAUVE_MAX_SEQ];
};

void measure(int N, RGap &r)
{
    for (int g = 0; g < N; g++) {
        int64_t lo, hi;
        if (r.a.st[g] > 0) { lo = r.a.st[g] + r.a.len; hi = r.b.st[g] - 1; }
        else { lo = -r.b.st[g] + r.b.len; hi = -r.a.st[g] - 1; }
        r.lo[g] = lo; r.len[g] = std::max<int64_t>(0, hi - lo + 1);
    }
}

}  // namespace

int recursive_anchoring(mauve_ctx *c, const mauve_params *p, int w0, std::vector<std::vector<HMatch>> &chains)
{
    const int N = c->nseq;
    const uint32_t full = N >= 32 ? 0xffffffffu : ((1u << N) - 1);
    std::vector<RGap> work;
    for (size_t l = 0; l < chains.size(); l++)
        for (size_t i = 0; i + 1 < chains[l].size(); i++) {
            RGap r; r.lcb = (int64_t)l; r.a = chains[l][i]; r.b = chains[l][i + 1]; r.prev_w = w0;
            work.push_back(r);
        }
    std::vector<std::vector<HMatch>> found(chains.size());

    while (!work.empty()) {
        std::map<int, std::vector<size_t>> classes;     // seed weight -> gaps of this level
        for (size_t i = 0; i < work.size(); i++) {
            RGap &r = work[i];
            measure(N, r);
            int64_t mx = 0, mn = -1, sum = 0;
            for (int g = 0; g < N; g++) { mx = std::max(mx, r.len[g]); mn = mn < 0 ? r.len[g] : std::min(mn, r.len[g]); sum += r.len[g]; }
            if (mx <= p->min_recursive_gap) continue;
            int w = mauve_default_seed_weight(sum / N);
            if (w > r.prev_w - 2) w = r.prev_w - 2;
            if (w < 5) continue;
            if (mn < mauve_seed_length(mauve_get_seed(w, 0))) continue;
            classes[w].push_back(i);
        }
        std::vector<RGap> next;
        for (auto &cls : classes) {
            const int w = cls.first;
            const std::vector<size_t> &ids = cls.second;
            const uint32_t K = (uint32_t)ids.size();
            const uint64_t pat = mauve_get_seed(w, 0);
            // ---- virtual genomes: per genome, the gap sub-sequences in LCB orientation, concatenated ----
            GenomeSet vs; vs.buf = &c->rec_genomes; vs.nseq = N; vs.lens.assign(N, 0); vs.word_off.assign(N, 0);
            std::vector<uint32_t> seg((size_t)N * (K + 1));
            std::vector<std::vector<uint64_t>> packed(N);
            size_t words = 0;
            for (int g = 0; g < N; g++) {
                int64_t tot = 0;
                for (uint32_t k = 0; k < K; k++) { seg[(size_t)g * (K + 1) + k] = (uint32_t)tot; tot += work[ids[k]].len[g]; }
                seg[(size_t)g * (K + 1) + K] = (uint32_t)tot;
                if (tot >= (1LL << 31)) { c->err = "recursive anchoring: gap set too large"; return MAUVE_ERR_LIMIT; }
                vs.lens[g] = tot;
                std::vector<uint8_t> codes((size_t)tot + 1);
                const auto &hw = c->host_packed[g];
                for (uint32_t k = 0; k < K; k++) {
                    const RGap &r = work[ids[k]];
                    uint8_t *out = codes.data() + seg[(size_t)g * (K + 1) + k];
                    const int64_t lo0 = r.lo[g] - 1, n = r.len[g];
                    if (r.a.st[g] > 0) for (int64_t i = 0; i < n; i++) out[i] = base_at(hw, lo0 + i);
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
            // ---- per-gap chaining of the N-way forward matches ----
            const uint32_t *seg0 = seg.data();
            int64_t i = 0;
            while (i < nm) {
                const int64_t s0 = c->match_start[(size_t)i * N];      // genome 0 is always forward
                uint32_t k = (uint32_t)(std::upper_bound(seg0, seg0 + K + 1, (uint32_t)(s0 - 1)) - seg0) - 1;
                std::vector<HMatch> loc;
                while (i < nm && (uint32_t)(c->match_start[(size_t)i * N] - 1) < seg0[k + 1]) {
                    bool fwd = true;
                    HMatch h; h.len = c->match_len[(size_t)i];
                    for (int g = 0; g < N; g++) {
                        int64_t s = c->match_start[(size_t)i * N + g];
                        if (s <= 0) { fwd = false; break; }
                        h.st[g] = s - seg[(size_t)g * (K + 1) + k];        // 1-based inside the gap
                    }
                    if (fwd) loc.push_back(h);
                    i++;
                }
                if (loc.empty()) continue;
                host_eliminate_overlaps(N, loc);
                std::vector<int64_t> ml; int64_t nl = 0;
                host_lcb_chain(N, loc, 0, true, ml, nl);
                const RGap &r = work[ids[k]];
                std::vector<HMatch> glob;
                for (size_t q = 0; q < loc.size(); q++) {
                    if (ml[q] < 0) continue;
                    HMatch x; x.len = loc[q].len;
                    for (int g = 0; g < N; g++) {
                        const int64_t s = loc[q].st[g];
                        if (r.a.st[g] > 0) x.st[g] = r.lo[g] + s - 1;
                        else { const int64_t hi = r.lo[g] + r.len[g] - 1; x.st[g] = -(hi - (s - 1) - x.len + 1); }
                    }
                    glob.push_back(x);
                }
                if (glob.empty()) continue;
                std::sort(glob.begin(), glob.end(), [](const HMatch &x, const HMatch &y) { return x.st[0] < y.st[0]; });
                for (size_t q = 0; q <= glob.size(); q++) {
                    RGap sub; sub.lcb = r.lcb; sub.prev_w = w;
                    sub.a = q == 0 ? r.a : glob[q - 1];
                    sub.b = q == glob.size() ? r.b : glob[q];
                    next.push_back(sub);
                }
                for (const HMatch &x : glob) found[(size_t)r.lcb].push_back(x);
            }
        }
        work.swap(next);
    }
    for (size_t l = 0; l < chains.size(); l++) {
        if (found[l].empty()) continue;
        chains[l].insert(chains[l].end(), found[l].begin(), found[l].end());
        std::sort(chains[l].begin(), chains[l].end(), [](const HMatch &x, const HMatch &y) { return x.st[0] < y.st[0]; });
    }
    return MAUVE_OK;
}
