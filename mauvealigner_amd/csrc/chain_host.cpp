// chain_host.cpp -- host side of the chaining stage: MultiplicityFilter, EliminateOverlaps and the greedy
// breakpoint elimination of Aligner::align [EXT] (call sites mauveAligner.cpp:596,600,698; helper usage
// projectAndStrip.cpp:110-112, toGrimmFormat.cpp:51-79, sortContigs.cpp:55-84).
//
// The elimination loop is sequential by nature and runs over the compact LCB
// graph: per-genome doubly linked lists of LCB nodes, an ordered set keyed by (weight, genome-0
// order) for the minimum, and local re-merging around each removed node.  Spec: DESIGN.md S5.
#include "common.hpp"
#include <algorithm>
#include <cstdlib>
#include <numeric>
#include <set>

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

void MatchVec::sort_by_start0()
{
    const size_t n = size();
    std::vector<uint64_t> key(n), tmp;
    for (size_t i = 0; i < n; i++) key[i] = ((uint64_t)std::llabs(st(i)[0]) << 32) | (uint64_t)i;
    sort_by_left(key, tmp);
    std::vector<int64_t> nd(d.size());
    for (size_t r = 0; r < n; r++) std::copy(rec((uint32_t)key[r]), rec((uint32_t)key[r]) + 1 + N, nd.begin() + r * (1 + N));
    d.swap(nd);
}

void host_eliminate_overlaps(MatchVec &m)
{
    const int N = m.N;
    const size_t n = m.size();
    if (n < 2) return;
    std::vector<uint8_t> alive(n, 1);
    std::vector<uint64_t> key, key2, tmp; key.reserve(n); key2.reserve(n);
    std::vector<int64_t> cf(n, 0), cl(n, 0), lenv(n), leftv(n);
    std::vector<int8_t> fwd(n);
    std::vector<uint32_t> touched;
    static const bool trace = getenv("MAUVE_TRACE") != nullptr;
    for (int g = 0; g < N; g++) {
        // order of the alive matches in genome g; later passes keep it (crops and deletions rarely disturb it,
        // sort_by_left re-sorts only when they do)
        key.clear();
        for (size_t i = 0; i < n; i++) {
            if (!alive[i]) continue;
            leftv[i] = std::llabs(m.st(i)[g]); lenv[i] = m.len(i); fwd[i] = m.st(i)[g] > 0;
            key.push_back(((uint64_t)leftv[i] << 32) | (uint64_t)i);
        }
        for (int pass = 0;; pass++) {
            const double tp0 = trace ? now_ms() : 0;
            sort_by_left(key, tmp);
            touched.clear();
            for (size_t r = 0; r + 1 < key.size(); r++) {
                const uint32_t A = (uint32_t)key[r], B = (uint32_t)key[r + 1];
                const int64_t ov = leftv[A] + lenv[A] - leftv[B];
                if (ov <= 0) continue;
                if (lenv[A] < lenv[B]) {              // A gives up its right side in g
                    if (!cf[A] && !cl[A]) touched.push_back(A);
                    int64_t &c = fwd[A] ? cl[A] : cf[A];
                    c = std::max(c, ov);
                } else {                               // B gives up its left side in g
                    if (!cf[B] && !cl[B]) touched.push_back(B);
                    int64_t &c = fwd[B] ? cf[B] : cl[B];
                    c = std::max(c, ov);
                }
            }
            if (trace) fprintf(stderr, "[trace] eliminate g=%d pass=%d n=%zu overlaps=%zu %.3f ms\n", g, pass, key.size(), touched.size(), now_ms() - tp0);
            if (touched.empty()) break;
            bool died = false;
            for (uint32_t i : touched) {
                const int64_t nl = m.len(i) - cf[i] - cl[i];
                if (nl <= 0) { alive[i] = 0; died = true; }
                else {
                    for (int c = 0; c < N; c++) {
                        if (m.st(i)[c] > 0) m.st(i)[c] += cf[i];
                        else m.st(i)[c] -= cl[i];
                    }
                    m.len(i) = nl;
                    leftv[i] = std::llabs(m.st(i)[g]); lenv[i] = nl;
                }
                cf[i] = 0; cl[i] = 0;
            }
            // refresh the keys of the touched matches in place, drop the dead ones
            if (died) {
                key2.clear();
                for (uint64_t kk : key) { const uint32_t i = (uint32_t)kk; if (alive[i]) key2.push_back(((uint64_t)leftv[i] << 32) | i); }
                key.swap(key2);
            } else {
                for (uint64_t &kk : key) { const uint32_t i = (uint32_t)kk; kk = ((uint64_t)leftv[i] << 32) | i; }
            }
        }
    }
    size_t k = 0;
    for (size_t i = 0; i < n; i++) if (alive[i]) m.move(k++, i);
    m.resize(k);
}

namespace {
struct Node {
    int64_t weight = 0;
    std::vector<int32_t> prev, next;    // per genome
    bool alive = true;
    int32_t merged_into = -1;
};
}

void host_lcb_chain(const MatchVec &m, int64_t min_weight, bool collinear, std::vector<int64_t> &match_lcb, int64_t &n_lcb)
{
    const int N = m.N;
    const size_t n = m.size();
    match_lcb.assign(n, -1);
    n_lcb = 0;
    if (n == 0) return;
    // per-genome order of the matches
    std::vector<std::vector<uint32_t>> order(N, std::vector<uint32_t>(n)), rank(N, std::vector<uint32_t>(n));
    {
        for (int g = 0; g < N; g++) {
            std::vector<uint64_t> key(n), tmp;
            for (size_t i = 0; i < n; i++) key[i] = ((uint64_t)std::llabs(m.st(i)[g]) << 32) | (uint64_t)i;
            sort_by_left(key, tmp);
            for (uint32_t r = 0; r < n; r++) { order[g][r] = (uint32_t)key[r]; rank[g][order[g][r]] = r; }
        }
    }
    // initial nodes: maximal collinear runs in genome-0 order
    std::vector<int32_t> node_of(n);
    std::vector<Node> nodes;
    std::vector<uint32_t> node_first;     // first match (genome-0 order) of each node
    for (uint32_t k = 0; k < n; k++) {
        uint32_t i = order[0][k];
        bool join = k > 0;
        if (join) {
            uint32_t p = order[0][k - 1];
            for (int g = 1; g < N && join; g++) {
                bool oi = m.st(i)[g] < 0, op = m.st(p)[g] < 0;
                if (oi != op) join = false;
                else if (!oi) join = rank[g][i] == rank[g][p] + 1;
                else join = rank[g][i] + 1 == rank[g][p];
            }
        }
        if (!join) { nodes.emplace_back(); nodes.back().prev.assign(N, -1); nodes.back().next.assign(N, -1); node_first.push_back(i); }
        node_of[i] = (int32_t)nodes.size() - 1;
        nodes.back().weight += m.len(i) * N;
    }
    const int32_t K = (int32_t)nodes.size();
    // per-genome linked lists of nodes
    for (int g = 0; g < N; g++) {
        int32_t last = -1;
        for (uint32_t r = 0; r < n; r++) {
            int32_t nd = node_of[order[g][r]];
            if (nd == last) continue;
            // a node's matches are contiguous in every genome, so each node shows up exactly once here
            nodes[nd].prev[g] = last;
            if (last >= 0) nodes[last].next[g] = nd;
            last = nd;
        }
    }
    auto orient = [&](int32_t nd, int g) { return m.st(node_first[nd])[g] < 0; };
    auto mergeable = [&](int32_t a, int32_t b) {   // b == next_0(a)
        for (int g = 1; g < N; g++) {
            bool oa = orient(a, g);
            if (oa != orient(b, g)) return false;
            if (!oa ? nodes[a].next[g] != b : nodes[a].prev[g] != b) return false;
        }
        return true;
    };
    auto unlink = [&](int32_t x, int g) {
        int32_t p = nodes[x].prev[g], q = nodes[x].next[g];
        if (p >= 0) nodes[p].next[g] = q;
        if (q >= 0) nodes[q].prev[g] = p;
    };
    std::set<std::pair<int64_t, int32_t>> heap;    // (weight, genome-0 order index): node ids are in that order
    for (int32_t i = 0; i < K; i++) heap.insert({nodes[i].weight, i});
    int32_t alive_cnt = K;
    while (!heap.empty()) {
        auto it = heap.begin();
        if (collinear ? alive_cnt <= 1 : it->first >= min_weight) break;
        int32_t x = it->second;
        heap.erase(it);
        // neighbours that may become mergeable once x is gone
        std::vector<std::pair<int32_t, int32_t>> cand;
        for (int g = 0; g < N; g++) cand.push_back({nodes[x].prev[g], nodes[x].next[g]});
        for (int g = 0; g < N; g++) unlink(x, g);
        nodes[x].alive = false; alive_cnt--;
        for (auto pr : cand) {
            int32_t a = pr.first, b = pr.second;
            auto resolve = [&](int32_t v) { while (v >= 0 && !nodes[v].alive && nodes[v].merged_into >= 0) v = nodes[v].merged_into; return v; };
            a = resolve(a); b = resolve(b);
            if (a < 0 || b < 0 || a == b || !nodes[a].alive || !nodes[b].alive) continue;
            if (nodes[b].next[0] == a) std::swap(a, b);
            if (nodes[a].next[0] != b) continue;
            if (!mergeable(a, b)) continue;
            heap.erase({nodes[a].weight, a}); heap.erase({nodes[b].weight, b});
            nodes[a].weight += nodes[b].weight;
            for (int g = 0; g < N; g++) unlink(b, g);
            nodes[b].alive = false; nodes[b].merged_into = a; alive_cnt--;
            heap.insert({nodes[a].weight, a});
        }
    }
    // final ids in genome-0 order
    std::vector<int64_t> final_id(K, -1);
    int64_t id = 0;
    for (int32_t i = 0; i < K; i++) if (nodes[i].alive) final_id[i] = id++;
    n_lcb = id;
    for (size_t i = 0; i < n; i++) {
        int32_t v = node_of[i];
        while (!nodes[v].alive && nodes[v].merged_into >= 0) v = nodes[v].merged_into;
        match_lcb[i] = nodes[v].alive ? final_id[v] : -1;
    }
}

extern "C" {

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
