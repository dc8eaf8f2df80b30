// libMems/GuideTree.h -- the guide tree as a NEWICK file, the form ProgressiveAligner::setInputGuideTreeFileName /
// setOutputGuideTreeFileName exchange it in (progressiveMauve.cpp:278-279, 689-692; mauveAligner.cpp:616-623 names the
// temporary "guide_tree" file the same way).  On the device side the tree is the table mauve_guide_tree returns:
// 2N-1 nodes, leaves 0..N-1, internal nodes N..2N-2 in merge order (children before parents, root last).
// Leaves are named seq1 .. seqN by position in the sequence table; a bare 1-based number is read as well.  Branch
// lengths are written from the UPGMA heights (distance in substitutions-free "coverage" units) and are
// ignored on input: only the topology orders the alignment.  A node with more than two children (an unrooted tree's
// trifurcating root) is resolved left to right.
#ifndef MAUVE_HIP_GUIDETREE_H
#define MAUVE_HIP_GUIDETREE_H

#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace mems {

// NEWICK text of a merge-order tree.  dist: [N*N] leaf distances in parts per million (may be empty: no lengths).
inline std::string guideTreeToNewick(int N, const std::vector<int32_t> &left, const std::vector<int32_t> &right,
                                     const std::vector<int64_t> &dist = std::vector<int64_t>())
{
    const int M = 2 * N - 1;
    std::vector<double> height((size_t)M, 0.0);
    const bool lengths = dist.size() == (size_t)N * N;
    if (lengths) {
        // average-linkage distances between the clusters, as the UPGMA that built the tree kept them
        std::vector<double> D((size_t)M * M, 0.0); std::vector<int64_t> size((size_t)M, 1);
        for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) D[(size_t)i * M + j] = (double)dist[(size_t)i * N + j] * 1e-6;
        std::vector<char> active((size_t)M, 0);
        for (int i = 0; i < N; i++) active[(size_t)i] = 1;
        for (int k = N; k < M; k++) {
            const int a = left[(size_t)k], b = right[(size_t)k];
            height[(size_t)k] = D[(size_t)a * M + b] / 2;
            if (height[(size_t)k] < height[(size_t)a]) height[(size_t)k] = height[(size_t)a];      // a caller's tree need not be ultrametric
            if (height[(size_t)k] < height[(size_t)b]) height[(size_t)k] = height[(size_t)b];
            size[(size_t)k] = size[(size_t)a] + size[(size_t)b];
            for (int x = 0; x < k; x++) if (active[(size_t)x] && x != a && x != b)
                D[(size_t)k * M + x] = D[(size_t)x * M + k] = ((double)size[(size_t)a] * D[(size_t)a * M + x] + (double)size[(size_t)b] * D[(size_t)b * M + x]) / (double)size[(size_t)k];
            active[(size_t)a] = active[(size_t)b] = 0; active[(size_t)k] = 1;
        }
    }
    // iterative post-order print (a caterpillar tree of thousands of genomes must not recurse that deep)
    std::string out;
    struct Frame { int node, stage, parent; };
    std::vector<Frame> stack; stack.push_back({M - 1, 0, -1});
    char buf[48];
    while (!stack.empty()) {
        Frame &f = stack.back();
        const int k = f.node;
        if (k < N) {
            out += "seq" + std::to_string(k + 1);
            if (lengths && f.parent >= 0) { snprintf(buf, sizeof buf, ":%.6f", height[(size_t)f.parent]); out += buf; }
            stack.pop_back();
            continue;
        }
        if (f.stage == 0) { f.stage = 1; out += '('; const int ch = left[(size_t)k]; stack.push_back({ch, 0, k}); }
        else if (f.stage == 1) { f.stage = 2; out += ','; const int ch = right[(size_t)k]; stack.push_back({ch, 0, k}); }
        else {
            out += ')';
            if (lengths && f.parent >= 0) { snprintf(buf, sizeof buf, ":%.6f", height[(size_t)f.parent] - height[(size_t)k]); out += buf; }
            stack.pop_back();
        }
    }
    out += ";\n";
    return out;
}

// NEWICK text -> merge-order table for N sequences.  false (and *err) when the text is not a tree over seq1..seqN.
inline bool guideTreeFromNewick(const std::string &text, int N, std::vector<int32_t> &left, std::vector<int32_t> &right, std::string *err = nullptr)
{
    auto fail = [&](const char *m) { if (err) *err = m; return false; };
    if (N < 2) return fail("at least two sequences required");
    const int M = 2 * N - 1;
    left.assign((size_t)M, -1); right.assign((size_t)M, -1);
    std::vector<char> seen((size_t)N, 0);
    int next = N;                                           // next internal id: children are numbered before their parent
    std::vector<std::vector<int>> open;                     // children collected so far, one entry per open parenthesis
    size_t i = 0; const size_t n = text.size();
    int root = -1; bool done = false;
    auto skip_ws = [&]() { while (i < n && isspace((unsigned char)text[i])) i++; };
    auto skip_length = [&]() {                              // ":0.123", "[comment]"
        for (;;) {
            skip_ws();
            if (i < n && text[i] == ':') { i++; skip_ws(); while (i < n && (isdigit((unsigned char)text[i]) || text[i] == '.' || text[i] == '-' || text[i] == '+' || text[i] == 'e' || text[i] == 'E')) i++; }
            else if (i < n && text[i] == '[') { while (i < n && text[i] != ']') i++; if (i < n) i++; }
            else break;
        }
    };
    auto read_label = [&]() {
        std::string s; skip_ws();
        if (i < n && (text[i] == '\'' || text[i] == '"')) { const char q = text[i++]; while (i < n && text[i] != q) s += text[i++]; if (i < n) i++; }
        else while (i < n && !strchr("(),:;[", text[i]) && !isspace((unsigned char)text[i])) s += text[i++];
        return s;
    };
    auto place = [&](int node) { if (open.empty()) root = node; else open.back().push_back(node); };
    while (!done) {
        skip_ws();
        if (i >= n) return fail("unterminated tree (no ';')");
        const char ch = text[i];
        if (ch == '(') { open.emplace_back(); i++; }
        else if (ch == ',') { i++; }
        else if (ch == ')') {
            i++;
            if (open.empty()) return fail("unbalanced ')'");
            std::vector<int> kids = open.back(); open.pop_back();
            (void)read_label(); skip_length();              // internal label and branch length: not used
            if (kids.empty()) return fail("empty group");
            int node = kids[0];
            for (size_t k = 1; k < kids.size(); k++) {      // more than two children: resolved left to right
                if (next >= M) return fail("more internal nodes than a tree over these sequences has");
                left[(size_t)next] = node; right[(size_t)next] = kids[k]; node = next++;
            }
            place(node);
        }
        else if (ch == ';') { i++; done = true; }
        else {
            std::string lab = read_label(); skip_length();
            if (lab.empty()) return fail("unexpected character");
            size_t d = 0;
            if (lab.size() > 3 && (lab[0] == 's' || lab[0] == 'S') && (lab[1] == 'e' || lab[1] == 'E') && (lab[2] == 'q' || lab[2] == 'Q')) d = 3;
            for (size_t k = d; k < lab.size(); k++) if (!isdigit((unsigned char)lab[k])) return fail("leaf names must be seq1..seqN (or 1..N)");
            const long id = atol(lab.c_str() + d);
            if (id < 1 || id > N) return fail("leaf number outside 1..N");
            if (seen[(size_t)(id - 1)]) return fail("a sequence appears twice");
            seen[(size_t)(id - 1)] = 1;
            place((int)(id - 1));
        }
    }
    if (!open.empty()) return fail("unbalanced '('");
    for (int g = 0; g < N; g++) if (!seen[(size_t)g]) return fail("a sequence is missing from the tree");
    if (root != M - 1 || next != M) return fail("not a single tree over all sequences");
    return true;
}

}  // namespace mems
#endif
