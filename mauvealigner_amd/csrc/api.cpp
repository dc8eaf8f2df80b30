// api.cpp -- context management and the seed-side entry points of the C-ABI (include/mauve_hip.h).
#include "common.hpp"
#include <dlfcn.h>
#include <algorithm>
#include <cstring>

static thread_local std::string g_create_err;

GenomeSet main_genome_set(mauve_ctx *c)
{
    GenomeSet gs;
    gs.buf = &c->genomes; gs.nseq = c->nseq; gs.lens = c->lens; gs.word_off = c->word_off;
    if (c->has_invalid) gs.vmask = &c->base_invalid;
    if (c->has_contigs) gs.cmask = &c->contig_mask;
    if (c->has_invalid || c->has_contigs) gs.mask_off = c->base_mask_off;
    return gs;
}

// tails of freshly uploaded genomes: the bits past the last base of genome blockIdx.x cleared, its padding words (and, behind the
// last genome, the buffer's four slack words) zeroed -- window reads beyond a genome's end are deterministic
struct GenomeTails { uint64_t off[MAUVE_MAX_SEQ]; int64_t len[MAUVE_MAX_SEQ]; uint64_t total_words; };
__global__ void genome_tail_fix(uint64_t *__restrict__ words, GenomeTails gt)
{
    const int g = blockIdx.x;
    const int64_t len = gt.len[g];
    const uint64_t data = (uint64_t)((len + 31) / 32);
    uint64_t *w = words + gt.off[g];
    const uint64_t end = (g + 1 == (int)gridDim.x ? gt.total_words + 4 : gt.off[g + 1]) - gt.off[g];      // words this genome owns (padding included)
    if (threadIdx.x == 0 && (len & 31)) w[data - 1] &= (1ULL << (2 * (len & 31))) - 1ULL;
    for (uint64_t k = data + threadIdx.x; k < end; k += 64) w[k] = 0;
}

// is p inside a page-locked (hipHostMalloc / hipHostRegister) allocation?  Plain pointers make the query fail: that error is swallowed.
bool host_pointer_is_pinned(const void *p)
{
    if (!p) return false;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

// the host copy of the packed genomes (the XMFA writer reads bases from it): made by the staged upload, or fetched back here
int host_genomes(mauve_ctx *c)
{
    if (c->host_copy_valid) return MAUVE_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, c->pin_genomes.ensure((c->total_words + 4) * sizeof(uint64_t)));
    HIPCHK(c, hipMemcpyAsync(c->pin_genomes.p, c->genomes.p, (c->total_words + 4) * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    uint64_t *stage = c->pin_genomes.as<uint64_t>();
    c->host_packed.assign((size_t)c->nseq, nullptr);
    for (int g = 0; g < c->nseq; g++) c->host_packed[(size_t)g] = stage + c->word_off[(size_t)g];
    c->host_copy_valid = true;
    return MAUVE_OK;
}

// ---- RCCL, resolved at run time (mauve_set_shard_rccl): the library does not link librccl, it uses the one the caller's communicator belongs to ----
namespace {
struct RcclApi {
    int (*all_gather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;      // ncclAllGather(send, recv, count, ncclDataType_t, ncclComm_t, stream)
    const char *(*error_string)(int) = nullptr;                                               // ncclGetErrorString
    bool ok = false;
};
RcclApi &rccl_api()
{
    static RcclApi api = []() {
        RcclApi a;
        void *sym = dlsym(RTLD_DEFAULT, "ncclAllGather");
        void *lib = nullptr;
        if (!sym) { lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL); if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL); if (lib) sym = dlsym(lib, "ncclAllGather"); }
        if (sym) {
            a.all_gather = reinterpret_cast<int (*)(const void *, void *, size_t, int, void *, hipStream_t)>(sym);
            void *es = lib ? dlsym(lib, "ncclGetErrorString") : dlsym(RTLD_DEFAULT, "ncclGetErrorString");
            a.error_string = reinterpret_cast<const char *(*)(int)>(es);
            a.ok = true;
        }
        return a;
    }();
    return api;
}
constexpr int NCCL_INT8 = 0, NCCL_INT64 = 4;          // ncclDataType_t (rccl.h: ncclInt8 = 0, ncclInt64 = 4)

// one exchange through RCCL: sizes (one int64 per rank), then the payloads padded to the largest; host -> device -> all-gather -> host
int shard_allgather_rccl(mauve_ctx *c, const void *send, size_t bytes, std::vector<std::pair<const char *, size_t>> &parts)
{
    RcclApi &api = rccl_api();
    const int W = c->shard_world;
    const double t0 = now_ms();
    HIPCHK(c, hipSetDevice(c->device));
    auto nccl = [&](int r, const char *what) {
        if (r == 0) return MAUVE_OK;
        c->err = std::string("shard (RCCL): ") + what + " failed: " + (api.error_string ? api.error_string(r) : std::to_string(r)); return MAUVE_ERR_HIP;
    };
    // sizes
    HIPCHK(c, c->shard_pin.ensure((size_t)(W + 1) * 8));
    HIPCHK(c, c->shard_dev.ensure((size_t)(W + 1) * 8 + 64));
    int64_t *hs = c->shard_pin.as<int64_t>(), *ds = c->shard_dev.as<int64_t>();
    hs[0] = (int64_t)bytes;
    HIPCHK(c, hipMemcpyAsync(ds, hs, 8, hipMemcpyHostToDevice, c->stream));
    { const int rc = nccl(api.all_gather(ds, ds + 1, 1, NCCL_INT64, c->shard_comm, c->stream), "ncclAllGather (sizes)"); if (rc) return rc; }
    HIPCHK(c, hipMemcpyAsync(hs + 1, ds + 1, (size_t)W * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<int64_t> sz(hs + 1, hs + 1 + W);
    int64_t mx = 16;
    for (int r = 0; r < W; r++) { if (sz[(size_t)r] < 0) { c->err = "shard (RCCL): negative size"; return MAUVE_ERR_STATE; } mx = std::max(mx, sz[(size_t)r]); }
    mx = (mx + 15) & ~(int64_t)15;
    // payloads, padded to the largest
    const size_t o_send = (((size_t)(W + 1) * 8) + 63) & ~(size_t)63, o_recv = o_send + (size_t)mx, total = o_recv + (size_t)W * mx;
    HIPCHK(c, c->shard_pin.ensure(total));
    HIPCHK(c, c->shard_dev.ensure(total + 64));
    char *hp = c->shard_pin.as<char>(), *dp = c->shard_dev.as<char>();
    if (bytes) memcpy(hp + o_send, send, bytes);
    if (bytes) HIPCHK(c, hipMemcpyAsync(dp + o_send, hp + o_send, bytes, hipMemcpyHostToDevice, c->stream));
    { const int rc = nccl(api.all_gather(dp + o_send, dp + o_recv, (size_t)mx, NCCL_INT8, c->shard_comm, c->stream), "ncclAllGather (payloads)"); if (rc) return rc; }
    HIPCHK(c, hipMemcpyAsync(hp + o_recv, dp + o_recv, (size_t)W * mx, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int r = 0; r < W; r++) { parts.push_back({hp + o_recv + (size_t)r * mx, (size_t)sz[(size_t)r]}); c->shard_stat.bytes_received += sz[(size_t)r]; }
    c->shard_stat.exchanges++; c->shard_stat.bytes_sent += (int64_t)bytes; c->shard_stat.ms += now_ms() - t0;
    return MAUVE_OK;
}
}  // namespace

int shard_allgather(mauve_ctx *c, const void *send, size_t bytes, std::vector<std::pair<const char *, size_t>> &parts)
{
    parts.clear();
    if (c->shard_comm && c->shard_on) return shard_allgather_rccl(c, send, bytes, parts);
    if (c->shard_world <= 1 || !c->shard_fn) { parts.push_back({static_cast<const char *>(send), bytes}); return MAUVE_OK; }
    const double t0 = now_ms();
    const void *recv = nullptr;
    std::vector<int64_t> sz((size_t)c->shard_world, 0);
    const int rc = c->shard_fn(c->shard_user, send, (int64_t)bytes, &recv, sz.data());
    if (rc || (!recv && bytes)) { c->err = "shard: the caller's all-gather failed (" + std::to_string(rc) + ")"; return MAUVE_ERR_STATE; }
    const char *p = static_cast<const char *>(recv);
    for (int r = 0; r < c->shard_world; r++) {
        if (sz[(size_t)r] < 0) { c->err = "shard: negative size from the all-gather"; return MAUVE_ERR_STATE; }
        parts.push_back({p, (size_t)sz[(size_t)r]}); p += sz[(size_t)r];
        c->shard_stat.bytes_received += sz[(size_t)r];
    }
    c->shard_stat.exchanges++; c->shard_stat.bytes_sent += (int64_t)bytes; c->shard_stat.ms += now_ms() - t0;
    return MAUVE_OK;
}

void shard_lpt(const std::vector<int64_t> &cost, int world, std::vector<int> &owner)
{
    const size_t n = cost.size();
    owner.assign(n, 0);
    if (world <= 1) return;
    std::vector<size_t> idx(n);
    for (size_t i = 0; i < n; i++) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return cost[a] > cost[b]; });
    std::vector<int64_t> load((size_t)world, 0);
    for (size_t i : idx) {
        int best = 0;
        for (int r = 1; r < world; r++) if (load[(size_t)r] < load[(size_t)best]) best = r;
        owner[i] = best; load[(size_t)best] += cost[i] > 0 ? cost[i] : 1;
    }
}

extern "C" {

// The second stream carries the one-wave launch of the DP (dp_step2: thousands of small workgroups) while the workgroup launches -- the tail of
// the stage -- run on the main stream (dp_launch_steps): lowest priority, so that where the dispatcher has a choice the large workgroups go first.
static hipError_t create_priority_stream(hipStream_t *out)
{
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); return hipStreamCreateWithFlags(out, hipStreamNonBlocking); }
    return hipStreamCreateWithPriority(out, hipStreamNonBlocking, least);
}

int mauve_ctx_create(int device, mauve_ctx **out)
{
    if (!out) return MAUVE_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_err = std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "count = 0") +
                       "); libmauve_hip has no CPU fallback";
        return MAUVE_ERR_NOGPU;
    }
    if (device < 0 || device >= ndev) { g_create_err = "device index out of range"; return MAUVE_ERR_ARG; }
    mauve_ctx *c = new (std::nothrow) mauve_ctx();
    if (!c) { g_create_err = "out of host memory"; return MAUVE_ERR_ARG; }
    c->device = device;
    if (const char *sp = getenv("MAUVE_SCHEDULE")) {            // A/B hook: how the host waits in hipStreamSynchronize (spin | yield | block)
        const unsigned f = !strcmp(sp, "spin") ? hipDeviceScheduleSpin : !strcmp(sp, "yield") ? hipDeviceScheduleYield : hipDeviceScheduleBlockingSync;
        (void)hipSetDevice(device); (void)hipSetDeviceFlags(f); (void)hipGetLastError();
    }
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = create_priority_stream(&c->stream2)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess) {
        g_create_err = std::string("HIP init failed: ") + hipGetErrorString(e);
        delete c;
        return MAUVE_ERR_HIP;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        snprintf(c->devname, sizeof c->devname, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
        if (prop.multiProcessorCount > 0) c->cus = prop.multiProcessorCount;
    }
    c->pool = new (std::nothrow) SpinPool(SpinPool::default_threads());
    if (!c->pool) { g_create_err = "out of host memory"; mauve_ctx_destroy(c); return MAUVE_ERR_ARG; }
    *out = c;
    return MAUVE_OK;
}

void mauve_ctx_destroy(mauve_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    delete c->pool; c->pool = nullptr;
    DevBuf *bufs[] = {&c->genomes, &c->keysA, &c->keysB, &c->valsA, &c->valsB, &c->hist, &c->totals, &c->posmask,
                      &c->hit_mask, &c->hit_pos, &c->hit_seg, &c->base_invalid, &c->contig_mask, &c->node_cmask, &c->run_sum, &c->join_ovf, &c->join_bound, &c->dpf_anch, &c->dpf_work, &c->dpf_tot, &c->ch_len, &c->ch_st, &c->ch_crop, &c->ch_ent, &c->ch_ord, &c->ch_rank, &c->ch_node, &c->ch_graph, &c->ch_cnt, &c->ch_anch, &c->ch_lw, &c->ch_anch2, &c->ext_work, &c->bp_work, &c->dp_sp, &c->hom_cols, &c->sorted_rec_keep, &c->ch_big, &c->gap_work, &c->as_wide, &c->res_narrow, &c->dp_pick, &c->dp_wflags, &c->ch_mw, &c->shard_dev, &c->as_work, &c->as_isl, &c->res_cols, &c->sorted_rec, &c->canon_k1, &c->canon_k2, &c->canon_v1, &c->canon_v2, &c->rec_genomes, &c->rec_seg, &c->rec_vinv, &c->rec_vcm, &c->placed_mask, &c->bb_cols, &c->bb_work, &c->bb_query, &c->cand, &c->mlen, &c->mstart, &c->counters, &c->dp_desc, &c->dp_list, &c->dp_codes, &c->dp_off,
                      &c->dp_prof_cnt, &c->dp_prof_mask, &c->dp_prof2_cnt, &c->dp_prof2_mask, &c->dp_tb, &c->dp_meta,
                      &c->dp_score, &c->dp_cols, &c->dp_rows};
    for (DevBuf *b : bufs) b->release();
    c->pin_genomes.release(); c->pin_tail.release(); c->pin_ext.release(); c->pin_chain.release(); c->pin_mask.release(); c->pin_bb.release(); c->pin_asm.release(); c->pin_tab.release(); c->pin_cols.release(); c->pin_anch.release(); c->pin_dcols.release(); c->pin_meta.release(); c->pin_seed.release(); c->pin_dp_in.release(); c->shard_pin.release();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *mauve_last_error(const mauve_ctx *c) { return c ? c->err.c_str() : g_create_err.c_str(); }

int mauve_device_name(const mauve_ctx *c, char *buf, size_t buflen)
{
    if (!c || !buf || !buflen) return MAUVE_ERR_ARG;
    snprintf(buf, buflen, "%s", c->devname);
    return MAUVE_OK;
}

int mauve_synchronize(mauve_ctx *c)
{
    if (!c) return MAUVE_ERR_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MAUVE_OK;
}

int mauve_set_shard(mauve_ctx *c, int rank, int world, mauve_allgather_fn fn, void *user)
{
    if (!c) return MAUVE_ERR_ARG;
    c->shard_comm = nullptr; c->shard_stat = mauve_shard_stats{0, 0, 0, 0.0};
    if (world <= 1 || !fn) { c->shard_rank = 0; c->shard_world = 1; c->shard_fn = nullptr; c->shard_user = nullptr; c->shard_on = false; return MAUVE_OK; }
    if (rank < 0 || rank >= world) { c->err = "set_shard: rank outside the world"; return MAUVE_ERR_ARG; }
    c->shard_rank = rank; c->shard_world = world; c->shard_fn = fn; c->shard_user = user; c->shard_on = true;
    return MAUVE_OK;
}

int mauve_set_shard_rccl(mauve_ctx *c, int rank, int world, void *nccl_comm)
{
    if (!c) return MAUVE_ERR_ARG;
    c->shard_fn = nullptr; c->shard_user = nullptr; c->shard_comm = nullptr; c->shard_on = false; c->shard_rank = 0; c->shard_world = 1;
    c->shard_stat = mauve_shard_stats{0, 0, 0, 0.0};
    if (world < 1 || !nccl_comm) return world <= 1 ? MAUVE_OK : (c->err = "set_shard_rccl: a communicator is required", MAUVE_ERR_ARG);
    if (rank < 0 || rank >= world) { c->err = "set_shard_rccl: rank outside the world"; return MAUVE_ERR_ARG; }
    if (!rccl_api().ok) { c->err = "set_shard_rccl: no RCCL in this process (ncclAllGather not found, librccl.so.1 not loadable)"; return MAUVE_ERR_STATE; }
    static const bool single = getenv("MAUVE_SHARD_SINGLE") != nullptr;
    if (world == 1 && !single) return MAUVE_OK;                 // one rank: nothing to deal out
    c->shard_rank = rank; c->shard_world = world; c->shard_comm = nccl_comm; c->shard_on = true;
    return MAUVE_OK;
}

int mauve_shard_get_stats(mauve_ctx *c, mauve_shard_stats *out)
{
    if (!c || !out) return MAUVE_ERR_ARG;
    *out = c->shard_stat;
    return MAUVE_OK;
}

int mauve_host_alloc(size_t bytes, void **out)
{
    if (!out) return MAUVE_ERR_ARG;
    *out = nullptr;
    if (hipHostMalloc(out, bytes ? bytes : 64, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return MAUVE_ERR_HIP; }
    return MAUVE_OK;
}

void mauve_host_free(void *p) { if (p) (void)hipHostFree(p); }

int mauve_set_genomes(mauve_ctx *c, int nseq, const uint64_t *const *packed, const int64_t *lens)
{
    if (!c) return MAUVE_ERR_ARG;
    if (nseq < 1 || nseq > MAUVE_MAX_SEQ || !packed || !lens) { c->err = "set_genomes: 1..32 sequences required"; return MAUVE_ERR_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    // A result still in the context was made from the genomes that are being replaced: it ends here (mauve_hip.h: results are held
    // until the next call).  What of it is still on the device is not brought over -- in a loop of set_genomes / align / fetch that
    // would copy every result twice -- and a later fetch is refused instead of handing out half a result.
    if (c->res.dev_pending || c->res.cols_pending) { c->res.dev_pending = false; c->res.cols_pending = false; c->res.stale = true; }
    c->res.genomes_replaced = true;       // what is on the host can still be fetched; mauve_apply_homology / mauve_write_xmfa on it are refused
    size_t total_words = 0;
    std::vector<uint64_t> off(nseq);
    int64_t total_len = 0;
    for (int g = 0; g < nseq; g++) {
        if (lens[g] < 0 || (lens[g] > 0 && !packed[g])) { c->err = "set_genomes: bad length or null pointer"; return MAUVE_ERR_ARG; }
        off[g] = total_words;
        total_words += mauve_packed_words(lens[g]);
        total_len += lens[g];
    }
    if (total_len >= (1LL << 31)) { c->err = "set_genomes: total length must stay below 2^31 bases"; return MAUVE_ERR_LIMIT; }
    HIPCHK(c, c->genomes.ensure((total_words + 4) * sizeof(uint64_t)));
    // Genomes in page-locked caller memory (mauve_host_alloc) go up straight from there: one DMA per genome, plus the few
    // words of its tail (the last data word with the bits past the last base cleared, the zero padding) from a small staging
    // block; the host copy that the XMFA writer reads is then fetched back from the device when it is first needed
    // (host_genomes).  Anything else is staged: the host copy lives in page-locked memory and doubles as the staging buffer
    // of ONE upload (a pageable source would go through the runtime's bounce buffers, fresh vectors per call cost page faults).
    bool all_pinned = true;
    for (int g = 0; g < nseq && all_pinned; g++) all_pinned = lens[g] == 0 || host_pointer_is_pinned(packed[g]);
    c->host_packed.assign((size_t)nseq, nullptr);
    uint64_t *dev = c->genomes.as<uint64_t>();
    if (all_pinned) {
        // one DMA when the caller's genomes lie one behind the other in the boundary's own layout (mauve_packed_words(len) words
        // each), else one per genome; the tails -- bits past the last base, the padding words -- are put right on the device
        bool contiguous = true;
        for (int g = 0; g + 1 < nseq && contiguous; g++) contiguous = packed[g + 1] == packed[g] + mauve_packed_words(lens[g]);
        if (contiguous) {
            const size_t body = total_words - mauve_packed_words(lens[nseq - 1]) + (size_t)((lens[nseq - 1] + 31) / 32);      // up to the last data word
            if (body) HIPCHK(c, hipMemcpyAsync(dev, packed[0], body * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        } else
            for (int g = 0; g < nseq; g++) {
                const size_t data = (size_t)((lens[g] + 31) / 32);
                if (data) HIPCHK(c, hipMemcpyAsync(dev + off[g], packed[g], data * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
            }
        GenomeTails gt; memset(&gt, 0, sizeof gt);
        for (int g = 0; g < nseq; g++) { gt.off[g] = off[g]; gt.len[g] = lens[g]; }
        gt.total_words = total_words;
        hipLaunchKernelGGL(genome_tail_fix, dim3((unsigned)nseq), dim3(64), 0, c->stream, dev, gt);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->host_copy_valid = false;
    } else {
        HIPCHK(c, c->pin_genomes.ensure((total_words + 4) * sizeof(uint64_t)));
        uint64_t *stage = c->pin_genomes.as<uint64_t>();
        for (int g = 0; g < nseq; g++) {
            const size_t nw = mauve_packed_words(lens[g]), data = (size_t)((lens[g] + 31) / 32);
            uint64_t *dst = stage + off[g];
            if (data) memcpy(dst, packed[g], data * sizeof(uint64_t));
            for (size_t k = data; k < nw; k++) dst[k] = 0;
            // clear any bits past the last base so window reads beyond the end are deterministic
            if (lens[g] & 31) dst[data - 1] &= (1ULL << (2 * (lens[g] & 31))) - 1ULL;
            c->host_packed[(size_t)g] = dst;
        }
        for (size_t k = total_words; k < total_words + 4; k++) stage[k] = 0;
        HIPCHK(c, hipMemcpyAsync(dev, stage, (total_words + 4) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->host_copy_valid = true;
    }
    c->total_words = total_words;
    c->nseq = nseq;
    c->lens.assign(lens, lens + nseq);
    c->word_off = off;
    c->has_invalid = false; c->has_contigs = false; c->h_invalid.clear(); c->h_contig.clear(); c->base_mask_off.clear();
    c->n_matches = 0; c->match_len.clear(); c->match_start.clear(); c->matches_pending = false;
    return MAUVE_OK;
}

void mauve_ambiguity_bitmap(const char *ascii, int64_t len, uint64_t *bits)
{
    const size_t words = (size_t)(len / 64 + 1);
    for (size_t w = 0; w < words; w++) bits[w] = 0;
    for (int64_t i = 0; i < len; i++) {
        switch (ascii[i]) {
        case 'A': case 'C': case 'G': case 'T': case 'a': case 'c': case 'g': case 't': break;
        default: bits[i >> 6] |= 1ULL << (i & 63);
        }
    }
}

int mauve_set_genomes_contigs(mauve_ctx *c, int nseq, const uint64_t *const *packed, const int64_t *lens, const int64_t *n_contigs,
                              const int64_t *contig_starts, const uint64_t *const *invalid)
{
    int rc = mauve_set_genomes(c, nseq, packed, lens);
    if (rc) return rc;
    // word layout of both bitmaps: genome after genome, (len + 63) / 64 + 2 words each (two words of slack: a window read
    // never leaves its genome's words)
    std::vector<uint64_t> off((size_t)nseq, 0);
    size_t words = 0;
    for (int g = 0; g < nseq; g++) { off[(size_t)g] = words; words += (size_t)((lens[g] + 63) / 64) + 2; }
    std::vector<uint64_t> inv(words, 0), cm(words, 0);
    bool any_inv = false, any_contig = false;
    size_t cs = 0;
    for (int g = 0; g < nseq; g++) {
        uint64_t *I = inv.data() + off[(size_t)g], *M = cm.data() + off[(size_t)g];
        if (invalid && invalid[g]) {
            const size_t w = (size_t)((lens[g] + 63) / 64);
            for (size_t k = 0; k < w; k++) { I[k] = invalid[g][k]; }
            if (lens[g] & 63) I[w - 1] &= (1ULL << (lens[g] & 63)) - 1ULL;
            for (size_t k = 0; k < w; k++) any_inv |= I[k] != 0;
        }
        const int64_t nc = n_contigs ? n_contigs[g] : 0;
        int64_t prev = -1;
        for (int64_t k = 0; k < nc; k++) {
            const int64_t b = contig_starts[cs + (size_t)k];
            if (b < 0 || b > lens[g] || (k == 0 && b != 0) || (k && b <= prev)) { c->err = "set_genomes_contigs: contig starts must ascend inside the sequence"; return MAUVE_ERR_ARG; }
            prev = b;
            if (b > 0 && b < lens[g]) { M[b >> 6] |= 1ULL << (b & 63); any_contig = true; }
        }
        cs += (size_t)nc;
    }
    if (!any_inv && !any_contig) return MAUVE_OK;
    if (any_inv) {
        HIPCHK(c, c->base_invalid.ensure(words * 8));
        HIPCHK(c, hipMemcpy(c->base_invalid.p, inv.data(), words * 8, hipMemcpyHostToDevice));
    }
    if (any_contig) {
        HIPCHK(c, c->contig_mask.ensure(words * 8));
        HIPCHK(c, hipMemcpy(c->contig_mask.p, cm.data(), words * 8, hipMemcpyHostToDevice));
    }
    c->has_invalid = any_inv; c->has_contigs = any_contig;
    c->base_mask_off = off;
    if (any_inv) c->h_invalid.swap(inv);
    if (any_contig) c->h_contig.swap(cm);
    return MAUVE_OK;
}

int mauve_seed_mums(mauve_ctx *c, uint64_t pattern, int mode, uint64_t mask, int extend, int64_t *n_matches)
{
    if (!c) return MAUVE_ERR_ARG;
    if (mode != MAUVE_MODE_MEM && mode != MAUVE_MODE_UNIQUE && mode != MAUVE_MODE_PAIRWISE) { c->err = "seed_mums: unknown mode"; return MAUVE_ERR_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    { int rcm = materialize_tables(c); if (rcm) return rcm; }           // the pass reuses the buffers a resident result lives in
    return seedpass_run(c, main_genome_set(c), pattern, mode, mask, extend, nullptr, 0, n_matches);
}

int mauve_extend_hits(mauve_ctx *c, uint64_t pattern, int64_t n_hits, const uint32_t *mask, const int64_t *pos, const uint8_t *strand,
                      int extend, int64_t *n_matches)
{
    if (!c) return MAUVE_ERR_ARG;
    if (n_hits < 0 || n_hits >= (1LL << 31) || (n_hits && (!mask || !pos || !strand))) { c->err = "extend_hits: bad argument"; return MAUVE_ERR_ARG; }
    if (c->nseq < 1) { c->err = "extend_hits: no genomes set"; return MAUVE_ERR_STATE; }
    HIPCHK(c, hipSetDevice(c->device));
    { int rcm = materialize_tables(c); if (rcm) return rcm; }
    const int N = c->nseq;
    const int span = mauve_seed_length(pattern);
    if (span < 1) { c->err = "extend_hits: bad seed pattern"; return MAUVE_ERR_ARG; }
    std::vector<uint32_t> off((size_t)N + 1, 0);                  // first global window index of every genome
    std::vector<int64_t> nwin((size_t)N, 0);
    for (int g = 0; g < N; g++) { nwin[(size_t)g] = std::max<int64_t>(0, c->lens[(size_t)g] - span + 1); off[(size_t)g + 1] = off[(size_t)g] + (uint32_t)nwin[(size_t)g]; }
    std::vector<uint32_t> rec((size_t)n_hits * (N + 1), 0);
    for (int64_t h = 0; h < n_hits; h++) {
        const uint32_t m = mask[h];
        if (__builtin_popcount(m) < 2 || (N < 32 && (m >> N))) { c->err = "extend_hits: a hit needs two or more components among the genomes set"; return MAUVE_ERR_ARG; }
        rec[(size_t)h * (N + 1)] = m;
        for (int g = 0; g < N; g++) {
            if (!(m >> g & 1)) continue;
            const int64_t p = pos[h * N + g];
            if (p < 0 || p >= nwin[(size_t)g]) { c->err = "extend_hits: window start outside its genome"; return MAUVE_ERR_ARG; }
            rec[(size_t)h * (N + 1) + 1 + g] = (off[(size_t)g] + (uint32_t)p) | (strand[h * N + g] ? 0x80000000u : 0u);
        }
    }
    HostHits hh; hh.n = (uint32_t)n_hits; hh.rec = rec.data();
    return seedpass_from_hits(c, main_genome_set(c), pattern, hh, extend, n_matches);
}

int mauve_get_matches(mauve_ctx *c, int64_t *length, int64_t *start)
{
    if (!c) return MAUVE_ERR_ARG;
    if (c->n_matches && (!length || !start)) { c->err = "get_matches: null output"; return MAUVE_ERR_ARG; }
    { int rcm = seed_matches_to_host(c); if (rcm) return rcm; }
    if (c->n_matches) {
        memcpy(length, c->match_len.data(), c->match_len.size() * sizeof(int64_t));
        memcpy(start, c->match_start.data(), c->match_start.size() * sizeof(int64_t));
    }
    return MAUVE_OK;
}

int mauve_sorted_mer_list(mauve_ctx *c, int seq, uint64_t pattern, uint64_t *mer_out, int64_t *pos_out, int64_t *n_out)
{
    if (!c || !n_out) return MAUVE_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    { int rcm = materialize_tables(c); if (rcm) return rcm; }
    std::vector<uint64_t> keys; std::vector<uint32_t> vals; int w = 0;
    int rc = seedpass_sorted_list(c, main_genome_set(c), seq, pattern, &keys, &vals, &w);
    if (rc) return rc;
    *n_out = (int64_t)keys.size();
    if (mer_out && pos_out)
        for (size_t i = 0; i < keys.size(); i++) {
            mer_out[i] = (keys[i] << (64 - 2 * w)) | (vals[i] >> 31);
            pos_out[i] = vals[i] & 0x7fffffffu;
        }
    return MAUVE_OK;
}

// SeedMatchEnumerator::FindMatches / HashMatch / SetDirection (SeedMatchEnumerator.h:19-33,71-141): the runs of the
// sorted mer list become matches on the device (seed_pass.hip: enum_runs / enum_write); only the CSR result travels.
int mauve_seed_match_enumerate(mauve_ctx *c, int seq, uint64_t pattern, int64_t min_multi, int64_t max_multi,
                               int direct_only, int64_t *n_out, int64_t *n_starts, int64_t *mult, int64_t *start_off,
                               int64_t *starts)
{
    if (!c || !n_out || !n_starts) return MAUVE_ERR_ARG;
    if ((mult || start_off || starts) && !(mult && start_off && starts)) { c->err = "seed_match_enumerate: mult, start_off and starts go together"; return MAUVE_ERR_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    { int rcm = materialize_tables(c); if (rcm) return rcm; }
    EnumRequest q; q.min_multi = min_multi; q.max_multi = max_multi; q.direct_only = direct_only != 0; q.n = q.ns = 0;
    q.mult = mult; q.start_off = start_off; q.starts = starts;
    int rc = seedpass_enumerate(c, main_genome_set(c), seq, pattern, q);
    if (rc) return rc;
    *n_out = q.n; *n_starts = q.ns;
    return MAUVE_OK;
}

int mauve_profile_enable(mauve_ctx *c, int on)
{
    if (!c) return MAUVE_ERR_ARG;
    c->prof = on != 0;
    return MAUVE_OK;
}

int mauve_profile_reset(mauve_ctx *c)
{
    if (!c) return MAUVE_ERR_ARG;
    for (int i = 0; i < MAUVE_K_COUNT; i++) { c->k_ms[i] = 0; c->k_launch[i] = 0; c->k_units[i] = 0; }
    return MAUVE_OK;
}

int mauve_profile_get(mauve_ctx *c, int kernel, double *total_ms, int64_t *launches, int64_t *units)
{
    if (!c || kernel < 0 || kernel >= MAUVE_K_COUNT) return MAUVE_ERR_ARG;
    if (total_ms) *total_ms = c->k_ms[kernel];
    if (launches) *launches = c->k_launch[kernel];
    if (units) *units = c->k_units[kernel];
    return MAUVE_OK;
}

int mauve_last_stage_times(mauve_ctx *c, mauve_stage_times *t)
{
    if (!c || !t) return MAUVE_ERR_ARG;
    *t = c->stage;
    return MAUVE_OK;
}

}  // extern "C"
