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

void host_eliminate_overlaps(int N, std::vector<HMatch> &m)
{
    const size_t n = m.size();
    if (n < 2) return;
    std::vector<uint8_t> alive(n, 1);
    std::vector<uint32_t> ord; ord.reserve(n);
    std::vector<int64_t> cf(n), cl(n);
    for (int g = 0; g < N; g++) {
        for (;;) {
            ord.clear();
            for (size_t i = 0; i < n; i++) if (alive[i]) ord.push_back((uint32_t)i);
            std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) {
                int64_t x = std::llabs(m[a].st[g]), y = std::llabs(m[b].st[g]);
                return x != y ? x < y : a < b;
            });
            bool any = false;
            for (size_t r = 0; r + 1 < ord.size(); r++) {
                uint32_t A = ord[r], B = ord[r + 1];
                int64_t ov = std::llabs(m[A].st[g]) + m[A].len - std::llabs(m[B].st[g]);
                if (ov <= 0) continue;
                if (!any) { std::fill(cf.begin(), cf.end(), 0); std::fill(cl.begin(), cl.end(), 0); any = true; }
                if (m[A].len < m[B].len) {            // A gives up its right side in g
                    int64_t &c = m[A].st[g] > 0 ? cl[A] : cf[A];
                    c = std::max(c, ov);
                } else {                               // B gives up its left side in g
                    int64_t &c = m[B].st[g] > 0 ? cf[B] : cl[B];
                    c = std::max(c, ov);
                }
            }
            if (!any) break;
            for (uint32_t i : ord) {
                if (!cf[i] && !cl[i]) continue;
                int64_t nl = m[i].len - cf[i] - cl[i];
                if (nl <= 0) { alive[i] = 0; continue; }
                for (int c = 0; c < N; c++) {
                    if (m[i].st[c] > 0) m[i].st[c] += cf[i];
                    else m[i].st[c] -= cl[i];
                }
                m[i].len = nl;
            }
        }
    }
    size_t k = 0;
    for (size_t i = 0; i < n; i++) if (alive[i]) m[k++] = m[i];
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

void host_lcb_chain(int N, const std::vector<HMatch> &m, int64_t min_weight, bool collinear,
                    std::vector<int64_t> &match_lcb, int64_t &n_lcb)
{
    const size_t n = m.size();
    match_lcb.assign(n, -1);
    n_lcb = 0;
    if (n == 0) return;
    // per-genome order of the matches
    std::vector<std::vector<uint32_t>> order(N, std::vector<uint32_t>(n)), rank(N, std::vector<uint32_t>(n));
    for (int g = 0; g < N; g++) {
        std::iota(order[g].begin(), order[g].end(), 0u);
        std::sort(order[g].begin(), order[g].end(), [&](uint32_t a, uint32_t b) {
            int64_t x = std::llabs(m[a].st[g]), y = std::llabs(m[b].st[g]);
            return x != y ? x < y : a < b;
        });
        for (uint32_t r = 0; r < n; r++) rank[g][order[g][r]] = r;
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
                bool oi = m[i].st[g] < 0, op = m[p].st[g] < 0;
                if (oi != op) join = false;
                else if (!oi) join = rank[g][i] == rank[g][p] + 1;
                else join = rank[g][i] + 1 == rank[g][p];
            }
        }
        if (!join) { nodes.emplace_back(); nodes.back().prev.assign(N, -1); nodes.back().next.assign(N, -1); node_first.push_back(i); }
        node_of[i] = (int32_t)nodes.size() - 1;
        nodes.back().weight += m[i].len * N;
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
    auto orient = [&](int32_t nd, int g) { return m[node_first[nd]].st[g] < 0; };
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
    std::vector<HMatch> m((size_t)*n_inout);
    for (size_t i = 0; i < m.size(); i++) {
        m[i].len = length[i];
        for (int g = 0; g < nseq; g++) { m[i].st[g] = start[i * nseq + g]; if (!m[i].st[g]) return MAUVE_ERR_ARG; }
    }
    host_eliminate_overlaps(nseq, m);
    for (size_t i = 0; i < m.size(); i++) {
        length[i] = m[i].len;
        for (int g = 0; g < nseq; g++) start[i * nseq + g] = m[i].st[g];
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
    std::vector<HMatch> m((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        m[i].len = length[i];
        for (int g = 0; g < N; g++) { m[i].st[g] = start[i * N + g]; if (!m[i].st[g]) return MAUVE_ERR_ARG; }
    }
    std::vector<int64_t> ml; int64_t K = 0;
    host_lcb_chain(N, m, min_weight, collinear != 0, ml, K);
    *n_lcb_out = K;
    if (match_lcb) std::copy(ml.begin(), ml.end(), match_lcb);
    if (left_end && right_end && weight) {
        for (int64_t i = 0; i < K * N; i++) { left_end[i] = 0; right_end[i] = 0; }
        for (int64_t i = 0; i < K; i++) weight[i] = 0;
        for (int64_t i = 0; i < n; i++) {
            int64_t l = ml[i]; if (l < 0) continue;
            weight[l] += m[i].len * N;
            for (int g = 0; g < N; g++) {
                int64_t s = m[i].st[g], le = std::llabs(s), re = le + m[i].len - 1;
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
