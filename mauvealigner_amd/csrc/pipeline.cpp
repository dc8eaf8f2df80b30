// pipeline.cpp -- host orchestration of the whole hot path behind mauve_align():
//   doAlignment (mauveAligner.cpp:70): MaskedMemHash multi-MUMs (:523-531,585) -> MultiplicityFilter /
//   EliminateOverlaps (:596-600) -> Aligner::align (:698): LCBs by greedy breakpoint elimination with
//   default weight 3*w*N (:648-653), recursive anchoring of gaps > min_recursive_gap_length
//   (:127,670-672), gapped alignment of every inter-anchor interval through the GappedAligner seam
//   (:674-676) -> addUnalignedIntervals (:748) -> IntervalList (WriteStandardAlignment :746-760).
// Device work: seed pass (seed_pass.hip) and batched DP (dp_batch.hip).  Host work: the sequential
// chaining over the compact LCB graph and the assembly of SoA results.  No oracle code is used.
#include "common.hpp"
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace {

inline uint8_t base_at(const std::vector<uint64_t> &w, int64_t i) { return (uint8_t)((w[(size_t)(i >> 5)] >> (2 * (i & 31))) & 3); }

struct Gap {            // inter-anchor interval in LCB orientation
    int64_t lo[MAUVE_MAX_SEQ];    // 1-based left end in genome coordinates
    int64_t len[MAUVE_MAX_SEQ];
    bool rev[MAUVE_MAX_SEQ];
};

void gap_between(int N, const HMatch &a, const HMatch &b, Gap &gp)
{
    for (int g = 0; g < N; g++) {
        int64_t lo, hi;
        if (a.st[g] > 0) { lo = a.st[g] + a.len; hi = b.st[g] - 1; gp.rev[g] = false; }
        else { lo = -b.st[g] + b.len; hi = -a.st[g] - 1; gp.rev[g] = true; }
        gp.lo[g] = lo; gp.len[g] = std::max<int64_t>(0, hi - lo + 1);
    }
}

void gap_codes(const mauve_ctx *c, const Gap &gp, int g, uint8_t *out)
{
    const auto &w = c->host_packed[g];
    const int64_t lo0 = gp.lo[g] - 1, n = gp.len[g];
    if (!gp.rev[g]) for (int64_t i = 0; i < n; i++) out[i] = base_at(w, lo0 + i);
    else for (int64_t i = 0; i < n; i++) out[i] = (uint8_t)(3 - base_at(w, lo0 + n - 1 - i));
}

}  // namespace

int recursive_anchoring(mauve_ctx *c, const mauve_params *p, int w0, std::vector<std::vector<HMatch>> &chains);

extern "C" {

int mauve_align(mauve_ctx *c, const mauve_params *p, mauve_align_sizes *sizes)
{
    if (!c || !p || !sizes) return MAUVE_ERR_ARG;
    if (c->nseq < 2) { c->err = "align: at least two genomes required"; return MAUVE_ERR_STATE; }
    HIPCHK(c, hipSetDevice(c->device));
    const int N = c->nseq;
    const double t0 = now_ms();
    AlignResult &R = c->res;
    R = AlignResult();
    memset(&c->stage, 0, sizeof c->stage);

    int64_t sum = 0; for (int g = 0; g < N; g++) sum += c->lens[g];
    int w = p->seed_weight > 0 ? p->seed_weight : mauve_default_seed_weight(sum / N);
    uint64_t pat = p->seed_pattern ? p->seed_pattern : mauve_get_seed(w, p->seed_rank);
    if (!pat) { c->err = "align: no seed pattern for this weight/rank"; return MAUVE_ERR_ARG; }
    w = mauve_seed_weight(pat);
    const uint32_t full = N >= 32 ? 0xffffffffu : ((1u << N) - 1);

    // ---- seed pass: N-way multi-MUMs (the multiplicity filter is pushed into the join) ----
    int64_t nm = 0;
    int rc = seedpass_run(c, main_genome_set(c), pat, p->mode, full, 1, nullptr, 0, &nm);
    if (rc) return rc;
    R.mum_length = c->match_len; R.mum_start = c->match_start;
    const double t1 = now_ms();
    c->stage.seed_ms = t1 - t0;

    // ---- chaining ----
    std::vector<HMatch> m((size_t)nm);
    for (int64_t i = 0; i < nm; i++) {
        m[i].len = c->match_len[i];
        for (int g = 0; g < N; g++) m[i].st[g] = c->match_start[(size_t)i * N + g];
    }
    host_eliminate_overlaps(N, m);
    const int64_t lcbw = p->lcb_weight >= 0 ? p->lcb_weight : (int64_t)3 * w * N;
    std::vector<int64_t> match_lcb; int64_t nl = 0;
    host_lcb_chain(N, m, lcbw, p->collinear != 0, match_lcb, nl);
    std::vector<std::vector<HMatch>> chains((size_t)nl);
    R.lcb_weight.assign((size_t)nl, 0);
    for (size_t i = 0; i < m.size(); i++) {
        int64_t l = match_lcb[i]; if (l < 0) continue;
        chains[(size_t)l].push_back(m[i]);            // m is sorted by genome-0 start (canonical order)
        R.lcb_weight[(size_t)l] += m[i].len * N;
    }
    const double t2 = now_ms();
    c->stage.chain_ms = t2 - t1;

    // ---- recursive anchoring ----
    if (p->recursive) {
        rc = recursive_anchoring(c, p, w, chains);
        if (rc) return rc;
    }
    const double t3 = now_ms();
    c->stage.recurse_ms = t3 - t2;

    // ---- gapped alignment of every inter-anchor interval ----
    struct GapRef { int64_t lcb, idx; Gap gp; bool dp; int64_t dp_slot; };
    std::vector<GapRef> gaps;
    std::vector<int64_t> seq_off; seq_off.push_back(0);
    int64_t n_dp = 0, code_total = 0;
    for (int64_t l = 0; l < nl; l++) {
        auto &ch = chains[(size_t)l];
        for (size_t i = 0; i + 1 < ch.size(); i++) {
            GapRef gr; gr.lcb = l; gr.idx = (int64_t)i; gr.dp = false; gr.dp_slot = -1;
            gap_between(N, ch[i], ch[i + 1], gr.gp);
            int64_t tot = 0, mx = 0; int nonempty = 0;
            for (int g = 0; g < N; g++) { tot += gr.gp.len[g]; mx = std::max(mx, gr.gp.len[g]); nonempty += gr.gp.len[g] > 0; }
            if (tot == 0) continue;
            if (p->gapped && nonempty >= 2 && mx <= p->max_gapped_len) {
                gr.dp = true; gr.dp_slot = n_dp++;
                for (int g = 0; g < N; g++) { code_total += gr.gp.len[g]; seq_off.push_back(code_total); }
            }
            gaps.push_back(gr);
        }
    }
    std::vector<uint8_t> codes((size_t)code_total + 1);
    for (const GapRef &gr : gaps) {
        if (!gr.dp) continue;
        for (int g = 0; g < N; g++) gap_codes(c, gr.gp, g, codes.data() + seq_off[(size_t)gr.dp_slot * N + g]);
    }
    std::vector<uint32_t> dcols((size_t)code_total + 1);
    std::vector<int64_t> dcol_off((size_t)n_dp + 1, 0), dscore((size_t)n_dp + 1, 0);
    int64_t cells = 0;
    rc = dp_batch_run(c, N, n_dp, codes.data(), seq_off.data(), &p->scoring, dcols.data(), dcol_off.data(), dscore.data(), &cells);
    if (rc) return rc;
    const double t4 = now_ms();
    c->stage.dp_ms = t4 - t3;

    // ---- assemble the interval table ----
    R.col_off.clear(); R.cols.clear();
    int64_t n_anchor = 0; for (auto &ch : chains) n_anchor += (int64_t)ch.size();
    R.anchor_length.reserve((size_t)n_anchor); R.anchor_start.reserve((size_t)n_anchor * N); R.anchor_lcb.reserve((size_t)n_anchor);
    R.lcb_left.assign((size_t)nl * N, 0); R.lcb_right.assign((size_t)nl * N, 0);
    R.dp_score.assign((size_t)nl, 0);
    size_t gi = 0;
    for (int64_t l = 0; l < nl; l++) {
        auto &ch = chains[(size_t)l];
        R.col_off.push_back((int64_t)R.cols.size());
        for (size_t i = 0; i < ch.size(); i++) {
            const HMatch &a = ch[i];
            R.anchor_length.push_back(a.len); R.anchor_lcb.push_back(l);
            for (int g = 0; g < N; g++) R.anchor_start.push_back(a.st[g]);
            R.cols.insert(R.cols.end(), (size_t)a.len, full);
            if (gi < gaps.size() && gaps[gi].lcb == l && gaps[gi].idx == (int64_t)i) {
                const GapRef &gr = gaps[gi++];
                if (gr.dp) {
                    R.cols.insert(R.cols.end(), dcols.begin() + dcol_off[(size_t)gr.dp_slot], dcols.begin() + dcol_off[(size_t)gr.dp_slot + 1]);
                    R.dp_score[(size_t)l] += dscore[(size_t)gr.dp_slot];
                } else {
                    for (int g = 0; g < N; g++) R.cols.insert(R.cols.end(), (size_t)gr.gp.len[g], 1u << g);
                }
            }
            for (int g = 0; g < N; g++) {
                int64_t le = std::llabs(a.st[g]), re = le + a.len - 1;
                int64_t &L = R.lcb_left[(size_t)l * N + g], &Rr = R.lcb_right[(size_t)l * N + g];
                if (L == 0 || le < std::llabs(L)) L = a.st[g] < 0 ? -le : le;
                if (Rr == 0 || re > std::llabs(Rr)) Rr = a.st[g] < 0 ? -re : re;
            }
        }
    }
    int64_t niv = nl;
    R.iv_left.assign((size_t)nl * N, 0); R.iv_right.assign((size_t)nl * N, 0); R.iv_reverse.assign((size_t)nl * N, 0);
    for (int64_t i = 0; i < nl * N; i++) {
        R.iv_left[(size_t)i] = std::llabs(R.lcb_left[(size_t)i]);
        R.iv_right[(size_t)i] = std::llabs(R.lcb_right[(size_t)i]);
        R.iv_reverse[(size_t)i] = R.lcb_left[(size_t)i] < 0;
    }
    if (p->add_unaligned) {
        for (int g = 0; g < N; g++) {
            std::vector<std::pair<int64_t, int64_t>> sp;
            for (int64_t l = 0; l < nl; l++) if (R.iv_left[(size_t)l * N + g]) sp.push_back({R.iv_left[(size_t)l * N + g], R.iv_right[(size_t)l * N + g]});
            std::sort(sp.begin(), sp.end());
            int64_t cur = 1;
            for (size_t i = 0; i <= sp.size(); i++) {
                int64_t lo = cur, hi = i < sp.size() ? sp[i].first - 1 : c->lens[g];
                if (hi >= lo) {
                    R.col_off.push_back((int64_t)R.cols.size());
                    R.cols.insert(R.cols.end(), (size_t)(hi - lo + 1), 1u << g);
                    for (int h = 0; h < N; h++) { R.iv_left.push_back(h == g ? lo : 0); R.iv_right.push_back(h == g ? hi : 0); R.iv_reverse.push_back(0); }
                    R.dp_score.push_back(0);
                    niv++;
                }
                if (i < sp.size() && sp[i].second + 1 > cur) cur = sp[i].second + 1;
            }
        }
    }
    R.col_off.push_back((int64_t)R.cols.size());
    R.sz.n_mums = nm; R.sz.n_lcb = nl; R.sz.n_anchor = n_anchor; R.sz.n_iv = niv; R.sz.n_cols = (int64_t)R.cols.size();
    R.sz.n_gap_dp = n_dp; R.sz.n_dp_cells = cells;
    *sizes = R.sz;
    const double t5 = now_ms();
    c->stage.assemble_ms = t5 - t4;
    c->stage.total_ms = t5 - t0;
    return MAUVE_OK;
}

#define CPY(dst, vec) do { if ((dst) && !(vec).empty()) memcpy((dst), (vec).data(), (vec).size() * sizeof((vec)[0])); } while (0)

int mauve_align_fetch(mauve_ctx *c, int64_t *mum_length, int64_t *mum_start, int64_t *lcb_left, int64_t *lcb_right,
                      int64_t *lcb_weight, int64_t *anchor_length, int64_t *anchor_start, int64_t *anchor_lcb,
                      int64_t *iv_left, int64_t *iv_right, int8_t *iv_reverse, int64_t *col_off, uint32_t *cols,
                      int64_t *dp_score)
{
    if (!c) return MAUVE_ERR_ARG;
    const AlignResult &R = c->res;
    CPY(mum_length, R.mum_length); CPY(mum_start, R.mum_start);
    CPY(lcb_left, R.lcb_left); CPY(lcb_right, R.lcb_right); CPY(lcb_weight, R.lcb_weight);
    CPY(anchor_length, R.anchor_length); CPY(anchor_start, R.anchor_start); CPY(anchor_lcb, R.anchor_lcb);
    CPY(iv_left, R.iv_left); CPY(iv_right, R.iv_right); CPY(iv_reverse, R.iv_reverse);
    CPY(col_off, R.col_off); CPY(cols, R.cols); CPY(dp_score, R.dp_score);
    return MAUVE_OK;
}

// IntervalList::WriteStandardAlignment [EXT] (mauveAligner.cpp:746-760); text format pinned by the in-tree
// writer mfa2xmfa.cpp:64 (header), :89-91 (#Sequence lines), :104-115 (entry line, 80-column rows, '=').
int mauve_write_xmfa(mauve_ctx *c, const char *const *names, char *buf, int64_t *len)
{
    if (!c || !len) return MAUVE_ERR_ARG;
    const AlignResult &R = c->res;
    const int N = c->nseq;
    static const char B[4] = {'A', 'C', 'G', 'T'};
    std::string out;
    out.reserve((size_t)R.cols.size() * (size_t)N / 2 + 4096);
    char line[600];
    out += "#FormatVersion Mauve1\n";
    for (int g = 0; g < N; g++) {
        snprintf(line, sizeof line, "#Sequence%dFile\t%s\n#Sequence%dEntry\t%d\n#Sequence%dFormat\tFastA\n", g + 1,
                 names && names[g] ? names[g] : "", g + 1, g + 1, g + 1);
        out += line;
    }
    std::string row;
    for (int64_t iv = 0; iv < R.sz.n_iv; iv++) {
        const int64_t c0 = R.col_off[(size_t)iv], nc = R.col_off[(size_t)iv + 1] - c0;
        for (int g = 0; g < N; g++) {
            const int64_t le = R.iv_left[(size_t)iv * N + g], re = R.iv_right[(size_t)iv * N + g];
            if (!le) continue;
            const bool rev = R.iv_reverse[(size_t)iv * N + g] != 0;
            int64_t nxt = rev ? re : le;
            row.resize((size_t)nc);
            const auto &w = c->host_packed[g];
            for (int64_t k = 0; k < nc; k++) {
                if (R.cols[(size_t)(c0 + k)] >> g & 1) {
                    uint8_t b = base_at(w, nxt - 1);
                    row[(size_t)k] = rev ? B[3 - b] : B[b];
                    nxt += rev ? -1 : 1;
                } else row[(size_t)k] = '-';
            }
            snprintf(line, sizeof line, "> %d:%lld-%lld %c %s\n", g + 1, (long long)le, (long long)re, rev ? '-' : '+',
                     names && names[g] ? names[g] : "");
            out += line;
            for (int64_t pos = 0; pos < nc; pos += 80) { out.append(row, (size_t)pos, (size_t)std::min<int64_t>(80, nc - pos)); out += '\n'; }
        }
        out += "=\n";
    }
    if (!buf) { *len = (int64_t)out.size() + 1; return MAUVE_OK; }
    if (*len < (int64_t)out.size() + 1) { c->err = "write_xmfa: buffer too small"; return MAUVE_ERR_ARG; }
    memcpy(buf, out.c_str(), out.size() + 1);
    *len = (int64_t)out.size() + 1;
    return MAUVE_OK;
}

}  // extern "C"
