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
#include <cstring>
#include <map>

namespace {

// Virtual genomes of one recursion batch, built on the device from the resident packed genomes: thread (g, j)
// writes packed word j of virtual genome g -- 32 bases, each looked up through the segment table (which gap it
// belongs to) and the gap's source range; reverse gaps are read backwards and complemented.  Pad words are zero.
struct RecGatherArgs {
    uint64_t src_word_off[MAUVE_MAX_SEQ];   // genome g's words inside the resident genome buffer
    uint64_t dst_word_off[MAUVE_MAX_SEQ];   // virtual genome g's words inside rec_genomes
    uint64_t dst_words[MAUVE_MAX_SEQ];      // including the 3 pad words
};

__global__ void __launch_bounds__(256) rec_gather(const uint64_t *__restrict__ genomes, uint64_t *__restrict__ out, RecGatherArgs ga,
                                                  const uint32_t *__restrict__ seg, const int64_t *__restrict__ glo0,
                                                  const uint8_t *__restrict__ grev, uint32_t K)
{
    const int g = blockIdx.y;
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= ga.dst_words[g]) return;
    const uint32_t *sg = seg + (size_t)g * (K + 1);
    const uint32_t tot = sg[K];
    const uint64_t base0 = j * 32;
    uint64_t word = 0;
    if (base0 < tot) {
        // last k with sg[k] <= base0: the (non-empty) gap that holds base0
        uint32_t lo = 0, hi = K;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (sg[mid] <= base0) lo = mid; else hi = mid; }
        uint32_t k = lo;
        const uint64_t *G = genomes + ga.src_word_off[g];
        for (int b = 0; b < 32; b++) {
            const uint64_t p = base0 + b;
            if (p >= tot) break;
            while (p >= sg[k + 1]) k++;                       // steps over empty gaps as well
            const int64_t off = (int64_t)(p - sg[k]), len = (int64_t)(sg[k + 1] - sg[k]), s0 = glo0[(size_t)g * K + k];
            const bool rev = grev[(size_t)g * K + k] != 0;
            const int64_t src = rev ? s0 + len - 1 - off : s0 + off;
            uint64_t code = (G[src >> 5] >> (2 * (src & 31))) & 3ULL;
            if (rev) code = 3ULL - code;
            word |= code << (2 * b);
        }
    }
    out[ga.dst_word_off[g] + j] = word;
}

// The unusable-base and contig-start bitmaps of the virtual genomes (only when the resident genomes carry them,
// mauve_set_genomes_contigs): thread (g, j) builds word j -- 64 virtual bases -- of both.  A contig boundary lies
// between a base and its predecessor, so on a reverse gap it moves to the other side of the base pair.
__global__ void __launch_bounds__(256) rec_gather_mask(const uint64_t *__restrict__ inv, const uint64_t *__restrict__ cm, uint64_t *__restrict__ vinv,
                                                       uint64_t *__restrict__ vcm, RecGatherArgs ga /* src = base mask words, dst = virtual mask words */,
                                                       const uint32_t *__restrict__ seg, const int64_t *__restrict__ glo0,
                                                       const uint8_t *__restrict__ grev, uint32_t K)
{
    const int g = blockIdx.y;
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= ga.dst_words[g]) return;
    const uint32_t *sg = seg + (size_t)g * (K + 1);
    const uint32_t tot = sg[K];
    const uint64_t base0 = j * 64;
    uint64_t wi = 0, wc = 0;
    if (base0 < tot) {
        uint32_t lo = 0, hi = K;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (sg[mid] <= base0) lo = mid; else hi = mid; }
        uint32_t k = lo;
        const uint64_t *I = inv ? inv + ga.src_word_off[g] : nullptr, *C = cm ? cm + ga.src_word_off[g] : nullptr;
        for (int b = 0; b < 64; b++) {
            const uint64_t p = base0 + b;
            if (p >= tot) break;
            while (p >= sg[k + 1]) k++;
            const int64_t off = (int64_t)(p - sg[k]), len = (int64_t)(sg[k + 1] - sg[k]), s0 = glo0[(size_t)g * K + k];
            const bool rev = grev[(size_t)g * K + k] != 0;
            const int64_t src = rev ? s0 + len - 1 - off : s0 + off;
            if (I && (I[src >> 6] >> (src & 63) & 1)) wi |= 1ULL << b;
            if (C && off > 0) {                                   // a boundary before the gap's first base is the gap's own edge
                const int64_t cb = rev ? src + 1 : src;           // the source base whose bit says "a contig starts here"
                if (C[cb >> 6] >> (cb & 63) & 1) wc |= 1ULL << b;
            }
        }
    }
    if (vinv) vinv[ga.dst_word_off[g] + j] = wi;
    if (vcm) vcm[ga.dst_word_off[g] + j] = wc;
}

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
    const double t_stage0 = now_ms();
    WorkList work(N);
    {
        // c->rec_flags (one byte per anchor in chain order, from the device: the gap behind it is longer than min_recursive_gap on some side):
        // consumed once; without it every gap is tested
        std::vector<uint8_t> flags; flags.swap(c->rec_flags);
        size_t total = 0; for (const MatchVec &ch : chains) total += ch.size();
        const bool use_flags = !flags.empty() && flags.size() == total;
        size_t off = 0;
        for (size_t l = 0; l < chains.size(); l++) {
            const size_t n = chains[l].size();
            if (use_flags) {
                const uint8_t *fl = flags.data() + off;
                for (size_t i = 0; i + 1 < n; i++)
                    if (fl[i] && gap_weight(N, chains[l].rec(i), chains[l].rec(i + 1), w0, p->min_recursive_gap))
                        work.push((int64_t)l, w0, chains[l].rec(i), chains[l].rec(i + 1));
            } else
                for (size_t i = 0; i + 1 < n; i++)
                    if (gap_weight(N, chains[l].rec(i), chains[l].rec(i + 1), w0, p->min_recursive_gap))
                        work.push((int64_t)l, w0, chains[l].rec(i), chains[l].rec(i + 1));
            off += n;
        }
    }
    std::vector<MatchVec> found(chains.size(), MatchVec(N));
    int level = 0;
    if (trace) fprintf(stderr, "[trace] recursion: work list of %zu gaps built in %.3f ms\n", work.size(), now_ms() - t_stage0);

    while (work.size()) {
        std::map<int, std::vector<size_t>> classes;     // seed weight -> gaps of this level
        for (size_t i = 0; i < work.size(); i++) {
            int w = gap_weight(N, work.a(i), work.b(i), work.prev_w(i), p->min_recursive_gap);
            if (w) classes[w].push_back(i);
        }
        WorkList next(N);
        next.d.reserve(work.d.size() + work.d.size() / 2);
        level++;
        for (auto &cls : classes) {
            const int w = cls.first;
            // Several contexts, one alignment (mauve_set_shard): the gaps of the batch are LPT-dealt by their size, this rank searches
            // and chains its share, and what every gap yields -- its new anchors -- is exchanged below, so that every rank goes on with
            // the same work list.  ids: this rank's gaps; pos_of[k]: their places in the whole batch (the order results are applied in).
            const std::vector<size_t> &ids_all = cls.second;
            std::vector<size_t> ids_mine; std::vector<uint32_t> pos_of;
            const bool sharded = c->shard_on && ids_all.size() >= (size_t)(2 * c->shard_world);
            if (sharded) {
                std::vector<int64_t> cost(ids_all.size(), 0);
                for (size_t q = 0; q < ids_all.size(); q++)
                    for (int g = 0; g < N; g++) { int64_t lo, ln; gap_of(work.a(ids_all[q]), work.b(ids_all[q]), g, lo, ln); cost[q] += ln; }
                std::vector<int> owner; shard_lpt(cost, c->shard_world, owner);
                for (size_t q = 0; q < ids_all.size(); q++) if (owner[q] == c->shard_rank) { ids_mine.push_back(ids_all[q]); pos_of.push_back((uint32_t)q); }
            }
            const std::vector<size_t> &ids = sharded ? ids_mine : ids_all;
            std::vector<int64_t> shard_msg;                  // per gap of this rank with new anchors: [place in the batch, n, n records]
            const uint32_t K = (uint32_t)ids.size();
            const double tc0 = now_ms();
            const uint64_t pat = mauve_get_seed(w, 0);
            // what a gap's new anchors (glob: real coordinates, genome-0 order) mean for the next level
            auto apply_gap = [&](size_t wi, const MatchVec &gl) {
                const int64_t lcb = work.lcb(wi);
                for (size_t q = 0; q <= gl.size(); q++)
                    next.push(lcb, w, q == 0 ? work.a(wi) : gl.rec(q - 1), q == gl.size() ? work.b(wi) : gl.rec(q));
                for (size_t q = 0; q < gl.size(); q++) found[(size_t)lcb].push(gl.rec(q));
            };
            // rc_local: what this rank's share of the batch came to.  A rank that failed still takes part in the exchange -- its message is the
            // marker [-1, status] -- so that the others do not wait in the collective for ever; every rank then returns an error.
            auto exchange = [&](int rc_local) -> int {
                if (!sharded) return rc_local;
                std::vector<std::pair<const char *, size_t>> parts;
                const std::string err_local = c->err;
                if (rc_local) { shard_msg.assign(2, 0); shard_msg[0] = -1; shard_msg[1] = rc_local; }
                int rcx = shard_allgather(c, shard_msg.data(), shard_msg.size() * 8, parts);
                if (rcx) return rcx;
                if (rc_local) { c->err = err_local; return rc_local; }
                // every rank's gaps, applied in the order of the whole batch
                std::vector<std::pair<uint32_t, const int64_t *>> got;
                for (size_t r = 0; r < parts.size(); r++) {
                    const auto &pt = parts[r];
                    const int64_t *v = reinterpret_cast<const int64_t *>(pt.first), *e = v + pt.second / 8;
                    if (pt.second % 8) { c->err = "recursion shard: a rank's part is not a whole number of words"; return MAUVE_ERR_STATE; }
                    if (e - v >= 2 && v[0] == -1) { c->err = "recursion shard: rank " + std::to_string(r) + " failed (status " + std::to_string(v[1]) + ")"; return MAUVE_ERR_STATE; }
                    while (v < e) {
                        if (e - v < 2 || v[0] < 0 || v[1] < 0 || v[1] > (e - v - 2) / (1 + N)) { c->err = "recursion shard: a rank's part ends inside a record"; return MAUVE_ERR_STATE; }
                        got.push_back({(uint32_t)v[0], v}); v += 2 + v[1] * (1 + N);
                    }
                }
                std::sort(got.begin(), got.end(), [](const std::pair<uint32_t, const int64_t *> &x, const std::pair<uint32_t, const int64_t *> &y) { return x.first < y.first; });
                MatchVec gl(N);
                for (const auto &gq : got) {
                    if (gq.first >= ids_all.size()) { c->err = "recursion shard: ranks disagree about the work list"; return MAUVE_ERR_STATE; }
                    gl.d.assign(gq.second + 2, gq.second + 2 + gq.second[1] * (1 + N));
                    apply_gap(ids_all[gq.first], gl);
                }
                return MAUVE_OK;
            };
            double tch0 = 0, t_elim = 0, t_lcb = 0;
            bool dev_chain = false, compact = false;
            auto body = [&]() -> int {                        // this rank's share of the batch; what it yields is applied (or put into shard_msg) by emit_gap
            if (K == 0) return MAUVE_OK;                      // (a rank without a gap of this class still takes part in the exchange)
            // ---- virtual genomes: per genome, the gap sub-sequences in LCB orientation, concatenated ----
            GenomeSet vs; vs.buf = &c->rec_genomes; vs.nseq = N; vs.lens.assign(N, 0); vs.word_off.assign(N, 0);
            std::vector<uint32_t> seg((size_t)N * (K + 1));
            std::vector<int64_t> glo((size_t)N * K), glen((size_t)N * K);
            std::vector<uint8_t> grev((size_t)N * K);
            RecGatherArgs ga; memset(&ga, 0, sizeof ga);
            size_t words = 0; uint64_t max_words = 0;
            for (int g = 0; g < N; g++) {
                int64_t tot = 0;
                for (uint32_t k = 0; k < K; k++) {
                    gap_of(work.a(ids[k]), work.b(ids[k]), g, glo[(size_t)g * K + k], glen[(size_t)g * K + k]);
                    seg[(size_t)g * (K + 1) + k] = (uint32_t)tot; tot += glen[(size_t)g * K + k];
                    grev[(size_t)g * K + k] = work.a(ids[k])[1 + g] > 0 ? 0 : 1;
                    glo[(size_t)g * K + k] -= 1;                  // 0-based for the gather; restored below
                }
                seg[(size_t)g * (K + 1) + K] = (uint32_t)tot;
                if (tot >= (1LL << 31)) { c->err = "recursive anchoring: gap set too large"; return MAUVE_ERR_LIMIT; }
                vs.lens[g] = tot;
                const size_t nw = mauve_packed_words(tot);
                vs.word_off[g] = words;
                ga.src_word_off[g] = c->word_off[gmap ? gmap[g] : g]; ga.dst_word_off[g] = words; ga.dst_words[g] = nw;
                max_words = std::max<uint64_t>(max_words, nw);
                words += nw;
            }
            HIPCHK(c, c->rec_genomes.ensure((words + 4) * sizeof(uint64_t)));
            // one side buffer: segment table, then the 0-based gap starts, then the strand flags
            const size_t seg_bytes = (seg.size() * sizeof(uint32_t) + 7) & ~(size_t)7, glo_bytes = glo.size() * sizeof(int64_t);
            HIPCHK(c, c->rec_seg.ensure(seg_bytes + glo_bytes + grev.size() + 8));
            char *side = c->rec_seg.as<char>();
            HIPCHK(c, hipMemcpyAsync(side, seg.data(), seg.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(side + seg_bytes, glo.data(), glo_bytes, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(side + seg_bytes + glo_bytes, grev.data(), grev.size(), hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(rec_gather, dim3((uint32_t)((max_words + 255) / 256), (uint32_t)N), dim3(256), 0, c->stream,
                               c->genomes.as<uint64_t>(), c->rec_genomes.as<uint64_t>(), ga, (const uint32_t *)side,
                               (const int64_t *)(side + seg_bytes), (const uint8_t *)(side + seg_bytes + glo_bytes), K);
            HIPCHK(c, hipGetLastError());
            if (c->has_invalid || c->has_contigs) {
                // the gaps inherit the ambiguity / contig bitmaps of the resident genomes
                RecGatherArgs gm; memset(&gm, 0, sizeof gm);
                vs.mask_off.assign((size_t)N, 0);
                size_t mwords = 0; uint64_t max_mw = 0;
                for (int g = 0; g < N; g++) {
                    const size_t nw = (size_t)((vs.lens[(size_t)g] + 63) / 64) + 2;
                    vs.mask_off[(size_t)g] = mwords; gm.src_word_off[g] = c->base_mask_off[(size_t)(gmap ? gmap[g] : g)]; gm.dst_word_off[g] = mwords; gm.dst_words[g] = nw;
                    max_mw = std::max<uint64_t>(max_mw, nw); mwords += nw;
                }
                HIPCHK(c, c->rec_vinv.ensure(mwords * 8)); HIPCHK(c, c->rec_vcm.ensure(mwords * 8));
                hipLaunchKernelGGL(rec_gather_mask, dim3((uint32_t)((max_mw + 255) / 256), (uint32_t)N), dim3(256), 0, c->stream,
                                   c->has_invalid ? c->base_invalid.as<uint64_t>() : nullptr, c->has_contigs ? c->contig_mask.as<uint64_t>() : nullptr,
                                   c->has_invalid ? c->rec_vinv.as<uint64_t>() : nullptr, c->has_contigs ? c->rec_vcm.as<uint64_t>() : nullptr, gm,
                                   (const uint32_t *)side, (const int64_t *)(side + seg_bytes), (const uint8_t *)(side + seg_bytes + glo_bytes), K);
                HIPCHK(c, hipGetLastError());
                if (c->has_invalid) vs.vmask = &c->rec_vinv;
                if (c->has_contigs) vs.cmask = &c->rec_vcm;
            }
            HIPCHK(c, hipStreamSynchronize(c->stream));    // the host vectors must outlive the copies
            for (auto &x : glo) x += 1;                    // back to 1-based for the coordinate mapping below
            int64_t nm = 0;
            static const bool host_gaps = getenv("MAUVE_HOST_GAP_CHAIN") != nullptr;      // A/B switch
            static const bool old_gaps = getenv("MAUVE_GAP_CHAIN_ALL_BACK") != nullptr;   // A/B switch: the device chain that copies every record and graph back
            c->lazy_matches_ok = !host_gaps && !old_gaps;     // a large batch is chained where its list is: no host copy of it
            int rc = seedpass_run(c, vs, pat, MAUVE_MODE_MEM, full, 1, c->rec_seg.as<uint32_t>(), K, &nm);
            c->lazy_matches_ok = false;
            if (rc) return rc;
            if (trace) fprintf(stderr, "[trace] recursion level %d weight %d: %u gaps, %lld bases, %lld matches, %.3f ms\n", level, w, K,
                               (long long)vs.lens[0], (long long)nm, now_ms() - tc0);
            // ---- per-gap chaining of the N-way forward matches ----
            tch0 = now_ms();
            const uint32_t *seg0 = seg.data();
            int64_t i = 0;
            MatchVec loc(N), glob(N);            // reused from gap to gap (thousands of gaps per batch)
            ChainOrders orders;
            std::vector<int64_t> ml;
            // Large batches are chained on the device, all gaps at once (chain_device_gaps); the host loop below then only
            // maps the survivors back.  Small batches (host-sorted lists: dev_rec_n != nm) and batches with an overlap cluster or a
            // gap sub-graph beyond the device kernels' limits (MAUVE_ERR_LIMIT) are chained here, gap by gap.  (A list with ties in
            // its canonical order is repaired by the seed pass before dev_rec_n is set, so it takes the device route like any
            // other.)  MAUVE_HOST_GAP_CHAIN: A/B switch.
            const int32_t *dl = nullptr, *ds = nullptr; const uint32_t *dg = nullptr; uint32_t ns = 0; std::vector<uint8_t> survive;
            if (!host_gaps && nm > 0 && c->dev_rec_n == nm) {
                int64_t maxlen = 1; for (int g = 0; g < N; g++) maxlen = std::max(maxlen, vs.lens[(size_t)g]);
                int rcg = old_gaps ? MAUVE_ERR_LIMIT : chain_device_gaps_compact(c, N, maxlen, c->rec_seg.as<uint32_t>(), K, &dl, &ds, &dg, &ns);
                if (rcg == MAUVE_OK) compact = true;
                else if (rcg != MAUVE_ERR_LIMIT) return rcg;
                else {
                    if (c->matches_pending) { rc = seed_matches_to_host(c); if (rc) return rc; }
                    rcg = chain_device_gaps(c, N, maxlen, c->rec_seg.as<uint32_t>(), K, &dl, &ds, survive);
                    if (rcg == MAUVE_OK) dev_chain = true;
                    else if (rcg != MAUVE_ERR_LIMIT) return rcg;
                }
            }
            if (c->matches_pending && !compact) { rc = seed_matches_to_host(c); if (rc) return rc; }
            // what a gap's surviving matches (loc: 1-based inside the gap, one collinear chain) mean for the next level
            auto emit_gap = [&](uint32_t k) {
                const size_t wi = ids[k];
                const int64_t *A = work.a(wi);
                glob.d.clear();
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
                if (glob.empty()) return;
                glob.sort_by_start0();
                if (sharded) {
                    shard_msg.push_back((int64_t)pos_of[k]); shard_msg.push_back((int64_t)glob.size());
                    shard_msg.insert(shard_msg.end(), glob.d.begin(), glob.d.end());
                } else apply_gap(wi, glob);
            };
            if (compact) {
                // only the survivors came back, with their gaps, in list order (gap by gap): straight to real coordinates
                for (uint32_t q = 0; q < ns;) {
                    const uint32_t k = dg[q];
                    uint32_t e = q + 1;
                    while (e < ns && dg[e] == k) e++;
                    const size_t wi = ids[k];
                    const int64_t *A = work.a(wi);
                    glob.d.resize((size_t)(e - q) * (1 + (size_t)N));
                    int64_t *o = glob.d.data();
                    for (; q < e; q++) {
                        const int64_t len = dl[q];
                        *o++ = len;
                        for (int g = 0; g < N; g++) {
                            const int64_t s = (int64_t)ds[(size_t)q * N + g] - seg[(size_t)g * (K + 1) + k], lo = glo[(size_t)g * K + k], ln = glen[(size_t)g * K + k];
                            *o++ = A[1 + g] > 0 ? lo + s - 1 : -((lo + ln - 1) - (s - 1) - len + 1);
                        }
                    }
                    glob.sort_by_start0();
                    if (sharded) {
                        shard_msg.push_back((int64_t)pos_of[k]); shard_msg.push_back((int64_t)glob.size());
                        shard_msg.insert(shard_msg.end(), glob.d.begin(), glob.d.end());
                    } else apply_gap(wi, glob);
                }
                i = nm;
            }
            while (i < nm) {
                const int64_t s0 = c->match_start[(size_t)i * N];      // genome 0 is always forward
                const uint32_t k = (uint32_t)(std::upper_bound(seg0, seg0 + K + 1, (uint32_t)(s0 - 1)) - seg0) - 1;
                loc.d.clear();
                while (i < nm && (uint32_t)(c->match_start[(size_t)i * N] - 1) < seg0[k + 1]) {
                    if (dev_chain) {                                   // the device's verdict and its cropped record
                        if (survive[(size_t)i]) {
                            int64_t rec[1 + MAUVE_MAX_SEQ]; rec[0] = dl[(size_t)i];
                            for (int g = 0; g < N; g++) rec[1 + g] = (int64_t)ds[(size_t)i * N + g] - seg[(size_t)g * (K + 1) + k];
                            loc.push(rec);
                        }
                        i++;
                        continue;
                    }
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
                if (dev_chain) ml.assign(loc.size(), 0);
                else {
                const double te0 = trace ? now_ms() : 0;
                host_eliminate_overlaps(loc, &orders);
                const double te1 = trace ? now_ms() : 0;
                int64_t nl = 0;
                host_lcb_chain(loc, 0, true, ml, nl, &orders);
                if (trace) { t_elim += te1 - te0; t_lcb += now_ms() - te1; }
                }
                emit_gap(k);
            }
            return MAUVE_OK;
            };
            { const int rc_body = body(); const int rcx = exchange(rc_body); if (rcx) return rcx; }
            if (trace) fprintf(stderr, "[trace]   per-gap chaining %.3f ms (%s; eliminate %.3f, lcb %.3f)\n", now_ms() - tch0, compact ? "device, survivors only" : (dev_chain ? "device" : "host"), t_elim, t_lcb);
        }
        work.d.swap(next.d);
    }
    const double tm0 = now_ms();
    // the new anchors into their chains: the chain is in genome-0 order already, so the (few) new ones are sorted and the two
    // lists merged -- distinct starts in genome 0, as anchors of one chain never overlap
    for (size_t l = 0; l < chains.size(); l++) {
        if (found[l].empty()) continue;
        found[l].sort_by_start0();
        // in place, from the back: the chain's buffer keeps its capacity from call to call, so no fresh 10 MB vector (and its page
        // faults) per alignment; every record moves once
        const size_t R1 = (size_t)(1 + N), na = chains[l].size(), nb = found[l].size();
        std::vector<int64_t> &A = chains[l].d;
        A.resize((na + nb) * R1);
        const int64_t *B = found[l].d.data();
        int64_t *D = A.data();
        size_t a = na, b = nb, o = na + nb;
        while (b > 0) {
            if (a > 0 && std::llabs(D[(a - 1) * R1 + 1]) > std::llabs(B[(b - 1) * R1 + 1])) {
                a--; o--;
                for (size_t k = 0; k < R1; k++) D[o * R1 + k] = D[a * R1 + k];
            } else {
                b--; o--;
                for (size_t k = 0; k < R1; k++) D[o * R1 + k] = B[b * R1 + k];
            }
        }                                                      // (what is left of A is in place already)
    }
    if (trace) fprintf(stderr, "[trace] recursion: merge into chains %.3f ms, whole stage %.3f ms\n", now_ms() - tm0, now_ms() - t_stage0);
    return MAUVE_OK;
}
