// chain_host.cpp -- host side of the chaining stage: MultiplicityFilter, EliminateOverlaps and the greedy
// breakpoint elimination of Aligner::align [EXT] (call sites mauveAligner.cpp:596,600,698; helper usage
// projectAndStrip.cpp:110-112, toGrimmFormat.cpp:51-79, sortContigs.cpp:55-84).
//
// The elimination loop is sequential by nature and runs over the compact LCB
// graph: flat per-genome doubly linked lists of LCB nodes, a lazy-deletion min-heap keyed by (weight,
// genome-0 order), and local re-merging around each removed node.  Spec: DESIGN.md S5.
#include "common.hpp"
#include <algorithm>
#include <cstdlib>
#include <numeric>

namespace {
// stable LSD radix sort of packed (left end << 32 | index) keys by the left end: 3 passes of 11 bits.
// Indices enter in ascending order, so ties keep index order -- the (left, index) order of the spec.
void sort_by_left(std::vector<uint64_t> &k, std::vector<uint64_t> &tmp)
{
    const size_t n = k.size();
    if (n < 2) return;
    bool sorted = true;
    for (size_t i = 1; i < n && sorted; i++) sorted = (k[i - 1] >> 32) <= (k[i] >> 32);
    if (sorted) return;
    tmp.resize(n);
    uint64_t *src = k.data(), *dst = tmp.data();
    for (int pass = 0; pass < 3; pass++) {
        const int sh = 32 + 11 * pass;
        uint32_t cnt[2049] = {0};
        for (size_t i = 0; i < n; i++) cnt[((src[i] >> sh) & 2047) + 1]++;
        for (int b = 0; b < 2048; b++) cnt[b + 1] += cnt[b];
        for (size_t i = 0; i < n; i++) dst[cnt[(src[i] >> sh) & 2047]++] = src[i];
        std::swap(src, dst);
    }
    if (src != k.data()) std::copy(src, src + n, k.data());
}
}  // namespace

// per-genome left-end order of an overlap-free list (what host_eliminate_overlaps would hand to host_lcb_chain)
void host_left_orders(const MatchVec &m, ChainOrders &orders)
{
    const int N = m.N; const size_t n = m.size();
    orders.ord.resize((size_t)N); orders.sparse = false;
    static thread_local std::vector<uint64_t> key, tmp;
    for (int g = 0; g < N; g++) {
        key.resize(n);
        for (size_t i = 0; i < n; i++) key[i] = ((uint64_t)std::llabs(m.st(i)[g]) << 32) | (uint64_t)i;
        sort_by_left(key, tmp);
        std::vector<uint32_t> &o = orders.ord[(size_t)g];
        o.resize(n);
        for (size_t r = 0; r < n; r++) o[r] = (uint32_t)key[r];
    }
}

// ---- seed families (progressiveMauve.cpp:502-546): the union of the searches of several seeds ----
// canonical order of match records (length, starts): first component, its start, component set, starts, length
static bool canon_less(int N, const int64_t *a, const int64_t *b)
{
    int fa = 0, fb = 0;
    while (fa < N && a[1 + fa] == 0) fa++;
    while (fb < N && b[1 + fb] == 0) fb++;
    if (fa != fb) return fa < fb;
    if (fa < N) { const int64_t sa = std::llabs(a[1 + fa]), sb = std::llabs(b[1 + fb]); if (sa != sb) return sa < sb; }
    uint32_t ma = 0, mb = 0;
    for (int g = 0; g < N; g++) { if (a[1 + g]) ma |= 1u << g; if (b[1 + g]) mb |= 1u << g; }
    if (ma != mb) return ma < mb;
    for (int g = 0; g < N; g++) if (a[1 + g] != b[1 + g]) return a[1 + g] < b[1 + g];
    return a[0] < b[0];
}
// x lies inside y: every component of x in y, same strand relation, same diagonal (y read in x's direction)
static bool match_contained(int N, const int64_t *x, const int64_t *y)
{
    const int64_t lx = x[0], ly = y[0];
    int f = 0;
    while (f < N && x[1 + f] == 0) f++;
    if (f == N || lx > ly || y[1 + f] == 0) return false;
    const bool flip = y[1 + f] < 0;
    int64_t d = -1;
    for (int g = f; g < N; g++) {
        const int64_t xs = x[1 + g];
        if (!xs) continue;
        if (!y[1 + g]) return false;
        const int64_t ys = flip ? -y[1 + g] : y[1 + g];
        if ((xs < 0) != (ys < 0)) return false;
        const int64_t ax = std::llabs(xs), ay = std::llabs(ys);
        int64_t dg = xs > 0 ? ax - ay : (ay + ly) - (ax + lx);
        if (flip) dg = (ly - lx) - dg;
        if (d < 0) d = dg;
        if (dg != d || dg < 0 || dg > ly - lx) return false;
    }
    return true;
}

// kept := kept, then the matches of add that no match of kept contains; canonical order.  For every genome g the kept
// matches that have g are ordered by their left end there with a running maximum of their right ends: the candidates
// that can hold a match x whose first component is g are the ones at or left of x's start, walked right to left until the
// running maximum falls short of x's right end.
void host_merge_matches(MatchVec &kept, const MatchVec &add)
{
    const int N = kept.N; const size_t R1 = (size_t)(1 + N);
    const size_t nk = kept.size(), na = add.size();
    if (!na) return;
    std::vector<std::vector<uint64_t>> by((size_t)N);          // per genome: (left end << 32 | index), ascending
    std::vector<std::vector<int64_t>> runmax((size_t)N);
    std::vector<char> need((size_t)N, 0);
    for (size_t j = 0; j < na; j++) { int f = 0; while (f < N && add.st(j)[f] == 0) f++; if (f < N) need[(size_t)f] = 1; }
    for (int g = 0; g < N; g++) {
        if (!need[(size_t)g]) continue;
        std::vector<uint64_t> &k = by[(size_t)g];
        k.reserve(nk);
        for (size_t i = 0; i < nk; i++) if (kept.st(i)[g]) k.push_back((uint64_t)std::llabs(kept.st(i)[g]) << 32 | (uint64_t)i);
        if (!std::is_sorted(k.begin(), k.end())) std::sort(k.begin(), k.end());       // canonical order is this order for the first component
        runmax[(size_t)g].resize(k.size());
        int64_t mx = 0;
        for (size_t r = 0; r < k.size(); r++) { const size_t i = (size_t)(uint32_t)k[r]; mx = std::max(mx, (int64_t)(k[r] >> 32) + kept.len(i) - 1); runmax[(size_t)g][r] = mx; }
    }
    std::vector<size_t> take;
    for (size_t j = 0; j < na; j++) {
        const int64_t *x = add.rec(j);
        int f = 0; while (f < N && x[1 + f] == 0) f++;
        bool dropped = false;
        if (f < N) {
            const std::vector<uint64_t> &k = by[(size_t)f];
            const int64_t xs = std::llabs(x[1 + f]), xe = xs + x[0] - 1;
            size_t r = (size_t)(std::upper_bound(k.begin(), k.end(), ((uint64_t)xs << 32) | 0xffffffffu) - k.begin());
            while (r > 0 && runmax[(size_t)f][r - 1] >= xe && !dropped) { r--; dropped = match_contained(N, x, kept.rec((size_t)(uint32_t)k[r])); }
        }
        if (!dropped) take.push_back(j);
    }
    if (take.empty()) return;
    // both lists are in canonical order: merge.  (Nearly every comparison is decided by the start in the first genome when both records have one
    // there -- the N-way lists of a seed family always do --; the full order only breaks the ties.  The buffer is kept from call to call: a fresh
    // 4 MB vector per merge cost its page faults.)
    static thread_local std::vector<int64_t> outbuf;
    outbuf.resize((nk + take.size()) * R1);
    auto less_fast = [&](const int64_t *x, const int64_t *y) {
        if (x[1] && y[1]) { const int64_t sx = std::llabs(x[1]), sy = std::llabs(y[1]); if (sx != sy) return sx < sy; }
        return canon_less(N, x, y);
    };
    size_t a = 0, b = 0, o = 0;
    while (a < nk || b < take.size()) {
        const bool ta = b >= take.size() || (a < nk && !less_fast(add.rec(take[b]), kept.rec(a)));
        const int64_t *src = ta ? kept.rec(a++) : add.rec(take[b++]);
        std::copy(src, src + R1, outbuf.begin() + (std::ptrdiff_t)(o++ * R1));
    }
    kept.d.swap(outbuf);
}

void MatchVec::sort_by_start0()
{
    const size_t n = size();
    {   // already in order (the usual case for lists that come out of a genome-0-ordered pass): nothing to do
        bool sorted = true;
        for (size_t i = 1; i < n && sorted; i++) sorted = std::llabs(st(i - 1)[0]) <= std::llabs(st(i)[0]);
        if (sorted) return;
    }
    static thread_local std::vector<uint64_t> key, tmp;
    static thread_local std::vector<int64_t> nd;        // swapped with d: both buffers live on, no fresh allocation
    key.resize(n);
    for (size_t i = 0; i < n; i++) key[i] = ((uint64_t)std::llabs(st(i)[0]) << 32) | (uint64_t)i;
    sort_by_left(key, tmp);
    nd.resize(d.size());
    for (size_t r = 0; r < n; r++) std::copy(rec((uint32_t)key[r]), rec((uint32_t)key[r]) + 1 + N, nd.begin() + r * (1 + N));
    d.swap(nd);
}

namespace {
// one alive match as genome g sees it; kept sorted by left end, so the overlap scan is a sequential sweep
struct Ent { uint32_t left, len, idx, flag; };     // flag: 1 = forward in g, 0 = reverse, 2 = died this pass

// order of the entries inside a genome: left end, ties by match index (every pass sorts by
// (left end, index) from scratch; equal left ends do occur after crops, mostly among very short matches)
inline bool ent_after(const Ent &a, const Ent &b) { return a.left != b.left ? a.left > b.left : a.idx > b.idx; }

// sort by (left end, index); returns at once when the order already holds
void sort_ents(std::vector<Ent> &e, std::vector<Ent> &tmp, bool nearly_sorted)
{
    const size_t n = e.size();
    if (n < 2) return;
    // Nearly sorted input (a pass of crops moves a few entries a few places) is repaired by an insertion sort on a
    // shift budget; whatever is left when the budget runs out goes through the radix passes.
    size_t budget = nearly_sorted ? 4 * n : 0, i = 1;
    for (; i < n; i++) {
        if (!ent_after(e[i - 1], e[i])) continue;
        const Ent x = e[i];
        size_t j = i;
        while (j > 0 && ent_after(e[j - 1], x) && budget) { e[j] = e[j - 1]; j--; budget--; }
        e[j] = x;
        if (!budget && j > 0 && ent_after(e[j - 1], x)) break;
    }
    if (i >= n) return;
    // stable LSD radix by left end (3 x 11 bits), then the (rare) groups of equal left ends by index
    tmp.resize(n);
    Ent *src = e.data(), *dst = tmp.data();
    for (int pass = 0; pass < 3; pass++) {
        const int sh = 11 * pass;
        uint32_t cnt[2049] = {0};
        for (size_t q = 0; q < n; q++) cnt[((src[q].left >> sh) & 2047) + 1]++;
        for (int b2 = 0; b2 < 2048; b2++) cnt[b2 + 1] += cnt[b2];
        for (size_t q = 0; q < n; q++) dst[cnt[(src[q].left >> sh) & 2047]++] = src[q];
        std::swap(src, dst);
    }
    if (src != e.data()) std::copy(src, src + n, e.data());
    for (size_t q = 0; q + 1 < n;) {
        size_t r = q + 1;
        while (r < n && e[r].left == e[q].left) r++;
        if (r - q > 1) std::sort(e.begin() + q, e.begin() + r, [](const Ent &x, const Ent &y) { return x.idx < y.idx; });
        q = r;
    }
}

// buffers kept per thread across calls: fresh multi-hundred-KB vectors cost page faults on every call
struct ElimScratch {
    std::vector<uint8_t> alive;
    std::vector<Ent> ents, tmp;
    std::vector<uint32_t> cf, cl, touched, newidx;
    std::vector<std::vector<uint32_t>> ordg;
};
}  // namespace

// After the crops the survivors are put back in canonical order (with three or more matches overlapping, the
// crops of one pass can carry a match past a neighbour; everything downstream reads the list as ordered along genome 0).
// alive == nullptr: every record is a survivor.  Dead records (sparse lists) go to the end.  orders are renumbered.
static void restore_canonical_order(MatchVec &m, const std::vector<uint8_t> *alive, ChainOrders *orders)
{
    const size_t n = m.size(); const int N = m.N;
    int64_t prev = 0; bool sorted = true;
    for (size_t i = 0; i < n && sorted; i++) {
        if (alive && !(*alive)[i]) continue;
        const int64_t s = std::llabs(m.st(i)[0]);
        sorted = s > prev; prev = s;
    }
    if (sorted) return;
    std::vector<uint32_t> perm; perm.reserve(n);
    for (size_t i = 0; i < n; i++) if (!alive || (*alive)[i]) perm.push_back((uint32_t)i);
    std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return std::llabs(m.st(a)[0]) < std::llabs(m.st(b)[0]); });
    if (alive) for (size_t i = 0; i < n; i++) if (!(*alive)[i]) perm.push_back((uint32_t)i);
    std::vector<uint32_t> newidx(n);
    MatchVec t(N); t.d.resize(m.d.size());
    const size_t R1 = (size_t)(1 + N);
    for (size_t k = 0; k < n; k++) { newidx[perm[k]] = (uint32_t)k; std::copy(m.rec(perm[k]), m.rec(perm[k]) + R1, t.d.begin() + (std::ptrdiff_t)(k * R1)); }
    m.d.swap(t.d);
    if (orders) for (auto &o : orders->ord) for (uint32_t &x : o) x = newidx[x];
}

// EliminateOverlaps.  Per genome: sweep the matches in left-end order; of two overlapping
// neighbours the shorter one gives up the overlap (ties: the right one), crops are collected per pass and
// applied together, and the sweep repeats until the genome is overlap free.  `orders` (optional) receives
// the per-genome order of the survivors -- the (left end, index) order host_lcb_chain needs: once a genome
// is overlap free, later crops and deaths cannot reorder it.
void host_eliminate_overlaps(MatchVec &m, ChainOrders *orders, bool compact)
{
    const int N = m.N;
    const size_t n = m.size();
    static const bool trace_on = getenv("MAUVE_TRACE") != nullptr;
    const bool trace = trace_on && n >= 1000;          // per-gap calls of the recursion stay quiet
    const double te0 = trace ? now_ms() : 0;
    static thread_local ElimScratch S;
    if (orders) { orders->ord.resize((size_t)N); for (auto &o : orders->ord) o.clear(); orders->sparse = false; }     // capacity is kept: called per gap
    if (n < 2) {
        if (orders) for (int g = 0; g < N; g++) orders->ord[(size_t)g].assign(n, 0u);
        return;
    }
    S.alive.assign(n, 1);
    if (S.cf.size() < n) { S.cf.assign(n, 0); S.cl.assign(n, 0); }     // all-zero between passes
    S.ordg.resize((size_t)N);
    std::vector<Ent> &ents = S.ents;
    double t_build = 0, t_ord = 0;
    if (trace) fprintf(stderr, "[trace] eliminate setup %.3f ms\n", now_ms() - te0);
    for (int g = 0; g < N; g++) {
        const double tb0 = trace ? now_ms() : 0;
        ents.clear();
        for (size_t i = 0; i < n; i++) {
            if (!S.alive[i]) continue;
            const int64_t s = m.st(i)[g];
            ents.push_back({(uint32_t)std::llabs(s), (uint32_t)m.len(i), (uint32_t)i, s > 0 ? 1u : 0u});
        }
        if (trace) t_build += now_ms() - tb0;
        for (int pass = 0;; pass++) {
            const double tp0 = trace ? now_ms() : 0;
            sort_ents(ents, S.tmp, pass > 0);
            S.touched.clear();
            const size_t k = ents.size();
            for (size_t r = 0; r + 1 < k; r++) {
                const Ent &A = ents[r], &B = ents[r + 1];
                const int64_t ov = (int64_t)A.left + A.len - B.left;
                if (ov <= 0) continue;
                if (A.len < B.len) {                  // A gives up its right side in g
                    if (!S.cf[r] && !S.cl[r]) S.touched.push_back((uint32_t)r);
                    uint32_t &c = A.flag ? S.cl[r] : S.cf[r];
                    c = std::max(c, (uint32_t)ov);
                } else {                               // B gives up its left side in g
                    if (!S.cf[r + 1] && !S.cl[r + 1]) S.touched.push_back((uint32_t)(r + 1));
                    uint32_t &c = B.flag ? S.cf[r + 1] : S.cl[r + 1];
                    c = std::max(c, (uint32_t)ov);
                }
            }
            if (trace) fprintf(stderr, "[trace] eliminate g=%d pass=%d n=%zu overlaps=%zu %.3f ms\n", g, pass, k, S.touched.size(), now_ms() - tp0);
            if (S.touched.empty()) break;
            bool died = false;
            for (uint32_t r : S.touched) {
                Ent &e = ents[r];
                const size_t i = e.idx;
                const int64_t cf = S.cf[r], cl = S.cl[r];
                const int64_t nl = m.len(i) - cf - cl;
                if (nl <= 0) { S.alive[i] = 0; e.flag = 2; died = true; }
                else {
                    for (int c = 0; c < N; c++) {
                        if (m.st(i)[c] > 0) m.st(i)[c] += cf;
                        else m.st(i)[c] -= cl;
                    }
                    m.len(i) = nl;
                    e.left = (uint32_t)std::llabs(m.st(i)[g]); e.len = (uint32_t)nl;
                }
                S.cf[r] = 0; S.cl[r] = 0;
            }
            if (died) {
                size_t w = 0;
                for (size_t r = 0; r < k; r++) if (ents[r].flag != 2) ents[w++] = ents[r];
                ents.resize(w);
            } else {
                // Nothing died: if no cropped entry moved past its right neighbour either, the genome is overlap free
                // (crops only shrink intervals and every overlapping adjacent pair was just resolved; an overlap
                // between non-neighbours would have pushed the entry between them out of order) -- the pass that
                // would merely confirm it is skipped.
                bool in_order = true;
                for (uint32_t r : S.touched) if (r + 1 < k && ent_after(ents[r], ents[r + 1])) { in_order = false; break; }
                if (in_order) break;
            }
        }
        const double to0 = trace ? now_ms() : 0;
        std::vector<uint32_t> &og = S.ordg[(size_t)g];
        og.resize(ents.size());
        for (size_t r = 0; r < ents.size(); r++) og[r] = ents[r].idx;
        if (trace) t_ord += now_ms() - to0;
    }
    if (trace) fprintf(stderr, "[trace] eliminate: entry build %.3f ms, order copies %.3f\n", t_build, t_ord);
    const double te1 = trace ? now_ms() : 0;
    if (!compact && orders) {
        // the list keeps its dead records; the survivors are exactly the entries of the order lists (old indices),
        // which is all host_lcb_chain needs -- saves moving every record and renumbering three index lists
        for (int g = 0; g < N; g++) {
            std::vector<uint32_t> &o = orders->ord[(size_t)g];
            o.clear();
            for (uint32_t i : S.ordg[(size_t)g]) if (S.alive[i]) o.push_back(i);
        }
        orders->sparse = true;
        restore_canonical_order(m, &S.alive, orders);
        if (trace) fprintf(stderr, "[trace] eliminate (no compaction) order lists %.3f ms, total %.3f\n", now_ms() - te1, now_ms() - te0);
        return;
    }
    if (orders) orders->sparse = false;
    S.newidx.resize(n);
    size_t k = 0;
    for (size_t i = 0; i < n; i++) if (S.alive[i]) { S.newidx[i] = (uint32_t)k; m.move(k++, i); }
    m.resize(k);
    if (orders) {
        for (int g = 0; g < N; g++) {
            std::vector<uint32_t> &o = orders->ord[(size_t)g];
            o.clear(); o.reserve(k);
            for (uint32_t i : S.ordg[(size_t)g]) if (S.alive[i]) o.push_back(S.newidx[i]);
        }
    }
    restore_canonical_order(m, nullptr, orders);
    if (trace) fprintf(stderr, "[trace] eliminate compaction %.3f ms, total %.3f\n", now_ms() - te1, now_ms() - te0);
}

// Greedy breakpoint elimination over the compact LCB graph: K nodes in genome-0 order with weights,
// per-genome doubly linked lists prevv/nextv [K][N] (-1 = none; modified), orient[nd] bit g = reverse in genome g.
// While the minimum weight is below min_weight (collinear: until one node is left) the minimum-weight node (first in
// genome-0 order on ties) is deleted and its neighbours re-merged.  final_id[nd] = id of the surviving LCB holding
// node nd, in genome-0 order, or -1.  Used by the host chain (below) and by the device chain (chain_dev.hip), which
// builds the same graph with kernels.
void lcb_greedy(int N, int32_t K, int64_t *weight, const uint32_t *orient_bits, int32_t *prevv, int32_t *nextv, int64_t min_weight,
                bool collinear, std::vector<int64_t> &final_id, int64_t &n_lcb)
{
    static thread_local std::vector<int32_t> merged_into;
    static thread_local std::vector<uint8_t> alive;
    merged_into.assign((size_t)K, -1);
    alive.assign((size_t)K, 1);
    auto PREV = [&](int32_t x, int g) -> int32_t & { return prevv[(size_t)x * N + g]; };
    auto NEXT = [&](int32_t x, int g) -> int32_t & { return nextv[(size_t)x * N + g]; };
    auto orient = [&](int32_t nd, int g) { return (orient_bits[(size_t)nd] >> g & 1u) != 0; };
    auto mergeable = [&](int32_t a, int32_t b) {   // b == next_0(a)
        for (int g = 1; g < N; g++) {
            const bool oa = orient(a, g);
            if (oa != orient(b, g)) return false;
            if (!oa ? NEXT(a, g) != b : PREV(a, g) != b) return false;
        }
        return true;
    };
    auto unlink = [&](int32_t x, int g) {
        const int32_t p = PREV(x, g), q = NEXT(x, g);
        if (p >= 0) NEXT(p, g) = q;
        if (q >= 0) PREV(q, g) = p;
    };
    // min-heap of (weight, node id) -- node ids are in genome-0 order, so ties resolve as the spec says -- with lazy
    // deletion: an entry counts only while its node is alive and still has the weight recorded in the entry
    // (weights only grow, by merges).  No per-node allocation: recursive anchoring calls this once per gap.
    typedef std::pair<int64_t, int32_t> HeapEnt;
    static thread_local std::vector<HeapEnt> heap;
    heap.clear();
    for (int32_t i = 0; i < K; i++) heap.push_back({weight[(size_t)i], i});
    auto cmp = [](const HeapEnt &a, const HeapEnt &b) { return a > b; };          // min-heap
    std::make_heap(heap.begin(), heap.end(), cmp);
    int32_t alive_cnt = K;
    static thread_local std::vector<std::pair<int32_t, int32_t>> cand;
    while (!heap.empty()) {
        const HeapEnt top = heap.front();
        if (!alive[(size_t)top.second] || weight[(size_t)top.second] != top.first) {     // stale entry
            std::pop_heap(heap.begin(), heap.end(), cmp); heap.pop_back();
            continue;
        }
        if (collinear ? alive_cnt <= 1 : top.first >= min_weight) break;
        const int32_t x = top.second;
        std::pop_heap(heap.begin(), heap.end(), cmp); heap.pop_back();
        // neighbours that may become mergeable once x is gone
        cand.clear();
        for (int g = 0; g < N; g++) cand.push_back({PREV(x, g), NEXT(x, g)});
        for (int g = 0; g < N; g++) unlink(x, g);
        alive[(size_t)x] = 0; alive_cnt--;
        for (auto pr : cand) {
            int32_t a = pr.first, b = pr.second;
            auto resolve = [&](int32_t v) { while (v >= 0 && !alive[(size_t)v] && merged_into[(size_t)v] >= 0) v = merged_into[(size_t)v]; return v; };
            a = resolve(a); b = resolve(b);
            if (a < 0 || b < 0 || a == b || !alive[(size_t)a] || !alive[(size_t)b]) continue;
            if (NEXT(b, 0) == a) std::swap(a, b);
            if (NEXT(a, 0) != b) continue;
            if (!mergeable(a, b)) continue;
            weight[(size_t)a] += weight[(size_t)b];              // the old entries of a and b are stale from here on
            for (int g = 0; g < N; g++) unlink(b, g);
            alive[(size_t)b] = 0; merged_into[(size_t)b] = a; alive_cnt--;
            heap.push_back({weight[(size_t)a], a}); std::push_heap(heap.begin(), heap.end(), cmp);
        }
    }
    // final ids in genome-0 order
    final_id.assign((size_t)K, -1);
    int64_t id = 0;
    for (int32_t i = 0; i < K; i++) {
        if (alive[(size_t)i]) final_id[(size_t)i] = id++;
    }
    n_lcb = id;
    // a dead node resolves through its merge chain; memoised per node, matches just look their node up
    for (int32_t i = 0; i < K; i++) {
        if (alive[(size_t)i]) continue;
        int32_t v = i;
        while (!alive[(size_t)v] && merged_into[(size_t)v] >= 0) v = merged_into[(size_t)v];
        final_id[(size_t)i] = alive[(size_t)v] ? final_id[(size_t)v] : -1;
    }
}

// Greedy breakpoint elimination over the compact LCB graph.  `orders` (optional): the
// per-genome (left end, index) order of m as produced by host_eliminate_overlaps; sorted here otherwise.
void host_lcb_chain(const MatchVec &m, int64_t min_weight, bool collinear, std::vector<int64_t> &match_lcb, int64_t &n_lcb,
                    const ChainOrders *orders, const int64_t *match_weight)
{
    const int N = m.N;
    const size_t n = m.size();
    match_lcb.assign(n, -1);
    n_lcb = 0;
    if (n == 0) return;
    static const bool trace_on = getenv("MAUVE_TRACE") != nullptr;
    const bool trace = trace_on && n >= 1000;
    const double tl0 = trace ? now_ms() : 0;
    // per-genome order of the matches and its inverse
    static thread_local std::vector<uint32_t> order_s, rank_s;
    static thread_local std::vector<uint64_t> key, tmp;
    order_s.resize((size_t)N * n); rank_s.resize((size_t)N * n);
    uint32_t *order = order_s.data(), *rank = rank_s.data();
    // `sparse` orders (host_eliminate_overlaps without compaction): m still holds dead records, the order lists name
    // the na survivors by their indices in m; everything below is indexed by those, dead records keep lcb -1
    const bool given = orders && (int)orders->ord.size() == N && (orders->sparse ? orders->ord[0].size() <= n : orders->ord[0].size() == n);
    const size_t na = given ? orders->ord[0].size() : n;
    if (na == 0) return;
    for (int g = 0; g < N; g++) {
        uint32_t *og = order + (size_t)g * n, *rg = rank + (size_t)g * n;
        if (given) std::copy(orders->ord[(size_t)g].begin(), orders->ord[(size_t)g].end(), og);
        else {
            key.resize(n);
            for (size_t i = 0; i < n; i++) key[i] = ((uint64_t)std::llabs(m.st(i)[g]) << 32) | (uint64_t)i;
            sort_by_left(key, tmp);
            for (size_t r = 0; r < n; r++) og[r] = (uint32_t)key[r];
        }
        for (uint32_t r = 0; r < na; r++) rg[og[r]] = r;
    }
    const double tl1 = trace ? now_ms() : 0;
    // initial nodes: maximal collinear runs in genome-0 order
    static thread_local std::vector<int32_t> node_of;
    node_of.resize(n);
    static thread_local std::vector<int64_t> weight;
    static thread_local std::vector<uint32_t> node_first;     // first match (genome-0 order) of each node
    weight.clear(); node_first.clear();
    for (uint32_t k = 0; k < na; k++) {
        const uint32_t i = order[k];
        bool join = k > 0;
        if (join) {
            const uint32_t p = order[k - 1];
            for (int g = 1; g < N && join; g++) {
                const uint32_t *rg = rank + (size_t)g * n;
                const bool oi = m.st(i)[g] < 0, op = m.st(p)[g] < 0;
                if (oi != op) join = false;
                else if (!oi) join = rg[i] == rg[p] + 1;
                else join = rg[i] + 1 == rg[p];
            }
        }
        if (!join) { weight.push_back(0); node_first.push_back(i); }
        node_of[i] = (int32_t)weight.size() - 1;
        weight.back() += match_weight ? match_weight[i] : m.len(i) * N;       // sum-of-pairs anchor scores instead
    }
    const int32_t K = (int32_t)weight.size();
    // per-genome doubly linked lists of nodes, flat [K][N]
    static thread_local std::vector<int32_t> prevv, nextv;
    static thread_local std::vector<uint32_t> orient;              // bit g: the node is reverse in genome g
    prevv.assign((size_t)K * N, -1); nextv.assign((size_t)K * N, -1); orient.assign((size_t)K, 0u);
    for (int32_t nd = 0; nd < K; nd++)
        for (int g = 0; g < N; g++) if (m.st(node_first[(size_t)nd])[g] < 0) orient[(size_t)nd] |= 1u << g;
    for (int g = 0; g < N; g++) {
        const uint32_t *og = order + (size_t)g * n;
        int32_t last = -1;
        for (uint32_t r = 0; r < na; r++) {
            const int32_t nd = node_of[og[r]];
            if (nd == last) continue;
            // a node's matches are contiguous in every genome, so each node shows up exactly once here
            prevv[(size_t)nd * N + g] = last;
            if (last >= 0) nextv[(size_t)last * N + g] = nd;
            last = nd;
        }
    }
    const double tl2 = trace ? now_ms() : 0;
    static thread_local std::vector<int64_t> final_id;
    lcb_greedy(N, K, weight.data(), orient.data(), prevv.data(), nextv.data(), min_weight, collinear, final_id, n_lcb);
    const double tl3 = trace ? now_ms() : 0;
    for (uint32_t k = 0; k < na; k++) { const uint32_t i = order[k]; match_lcb[i] = final_id[(size_t)node_of[i]]; }
    if (trace) fprintf(stderr, "[trace] lcb: orders %.3f ms, nodes+lists %.3f (K=%d), greedy %.3f, labels %.3f\n", tl1 - tl0, tl2 - tl1, K, tl3 - tl2, now_ms() - tl3);
}

extern "C" {

int mauve_merge_matches(int nseq, int64_t n_a, const int64_t *len_a, const int64_t *start_a, int64_t n_b, const int64_t *len_b, const int64_t *start_b,
                        int64_t *n_out, int64_t *len_out, int64_t *start_out)
{
    if (nseq < 1 || nseq > MAUVE_MAX_SEQ || n_a < 0 || n_b < 0 || !n_out) return MAUVE_ERR_ARG;
    MatchVec a(nseq), b(nseq);
    for (int64_t i = 0; i < n_a; i++) a.push(len_a[i], start_a + i * nseq);
    for (int64_t i = 0; i < n_b; i++) b.push(len_b[i], start_b + i * nseq);
    // the lists are taken in canonical order; a caller's list that is not gets sorted first
    auto canon = [&](MatchVec &m) {
        std::vector<size_t> idx(m.size()); std::iota(idx.begin(), idx.end(), 0);
        if (std::is_sorted(idx.begin(), idx.end(), [&](size_t x, size_t y) { return canon_less(nseq, m.rec(x), m.rec(y)); })) return;
        std::sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return canon_less(nseq, m.rec(x), m.rec(y)); });
        MatchVec t(nseq); for (size_t i : idx) t.push(m.rec(i)); m.d.swap(t.d);
    };
    canon(a); canon(b);
    host_merge_matches(a, b);
    if (len_out && start_out) {
        if (*n_out < (int64_t)a.size()) { *n_out = (int64_t)a.size(); return MAUVE_ERR_LIMIT; }
        for (size_t i = 0; i < a.size(); i++) { len_out[i] = a.len(i); std::copy(a.st(i), a.st(i) + nseq, start_out + i * nseq); }
    }
    *n_out = (int64_t)a.size();
    return MAUVE_OK;
}

int mauve_eliminate_overlaps(int nseq, int64_t *n_inout, int64_t *length, int64_t *start)
{
    if (nseq < 1 || nseq > MAUVE_MAX_SEQ || !n_inout || *n_inout < 0) return MAUVE_ERR_ARG;
    MatchVec m(nseq); m.resize((size_t)*n_inout);
    for (size_t i = 0; i < m.size(); i++) {
        m.len(i) = length[i];
        for (int g = 0; g < nseq; g++) { m.st(i)[g] = start[i * nseq + g]; if (!m.st(i)[g]) return MAUVE_ERR_ARG; }
    }
    host_eliminate_overlaps(m);
    for (size_t i = 0; i < m.size(); i++) {
        length[i] = m.len(i);
        for (int g = 0; g < nseq; g++) start[i * nseq + g] = m.st(i)[g];
    }
    *n_inout = (int64_t)m.size();
    return MAUVE_OK;
}

int mauve_lcb_chain(int nseq, int64_t n, const int64_t *length, const int64_t *start, int64_t min_weight, int collinear,
                    int64_t *match_lcb, int64_t *n_lcb_out, int64_t *left_end, int64_t *right_end, int64_t *weight,
                    int64_t *left_adj, int64_t *right_adj)
{
    if (nseq < 1 || nseq > MAUVE_MAX_SEQ || n < 0 || !n_lcb_out) return MAUVE_ERR_ARG;
    const int N = nseq;
    MatchVec m(N); m.resize((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        m.len(i) = length[i];
        for (int g = 0; g < N; g++) { m.st(i)[g] = start[i * N + g]; if (!m.st(i)[g]) return MAUVE_ERR_ARG; }
    }
    std::vector<int64_t> ml; int64_t K = 0;
    host_lcb_chain(m, min_weight, collinear != 0, ml, K);
    *n_lcb_out = K;
    if (match_lcb) std::copy(ml.begin(), ml.end(), match_lcb);
    if (left_end && right_end && weight) {
        for (int64_t i = 0; i < K * N; i++) { left_end[i] = 0; right_end[i] = 0; }
        for (int64_t i = 0; i < K; i++) weight[i] = 0;
        for (int64_t i = 0; i < n; i++) {
            int64_t l = ml[i]; if (l < 0) continue;
            weight[l] += m.len(i) * N;
            for (int g = 0; g < N; g++) {
                int64_t s = m.st(i)[g], le = std::llabs(s), re = le + m.len(i) - 1;
                int64_t &L = left_end[l * N + g], &R = right_end[l * N + g];
                if (L == 0 || le < std::llabs(L)) L = s < 0 ? -le : le;
                if (R == 0 || re > std::llabs(R)) R = s < 0 ? -re : re;
            }
        }
    }
    if (left_adj && right_adj && left_end) {
        // computeLCBAdjacencies_v2 semantics (toGrimmFormat.cpp:62-77): neighbours by left end per genome, -1 = none
        std::vector<int64_t> idx((size_t)K);
        for (int g = 0; g < N; g++) {
            std::iota(idx.begin(), idx.end(), 0);
            std::sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) { return std::llabs(left_end[a * N + g]) < std::llabs(left_end[b * N + g]); });
            for (int64_t r = 0; r < K; r++) {
                left_adj[idx[r] * N + g] = r > 0 ? idx[r - 1] : -1;
                right_adj[idx[r] * N + g] = r + 1 < K ? idx[r + 1] : -1;
            }
        }
    }
    return MAUVE_OK;
}

}  // extern "C"
