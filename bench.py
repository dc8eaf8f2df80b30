#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native mauveAligner hot path.

Metric (BASELINE.json): aligned Mbp/s = total bases / t(seed + extend + chain + recursive anchoring + gapped
DP), "t measured from packed genomes resident in host RAM to final SoA results in host RAM (includes H2D/D2H, excludes
FASTA parse and XMFA text formatting)".  A "step" is therefore: upload of the packed genomes (mauve_set_genomes), one
pass of the whole hot path (mauve_align / mauve_progressive_align) and the copy of every result array into the caller's
buffers (mauve_align_fetch).  The caller's buffers are page-locked (mauve_host_alloc), as a production caller's would be.
`value` is that figure; the same pass with the genomes already resident in HBM and the results left there is reported
beside it as `device_resident` (never as `value`).

Workload at N = 1: BASELINE config C3 -- 5 x 5 Mbp genomes, ~3 % divergence, ~50 inversions, seed weight 15: the
configuration the >= 50 Mbp/s target of `north_star` is quoted on.  The other BASELINE configs are measured in the same
run and reported as objects of their own ("c2", "c4", "c5"), each with its stage times, roofline and CPU baseline:
C2 (3 x 5 Mbp), C4 (8 x 2 Mbp, progressive path), C5 (2 x 100 Mbp, default seed weight 19, 64-bit keys).
--config picks another primary workload; --only-primary skips the others.

--gpus N, two forms (--shard):
  replicas (default): every rank aligns its own genome set of the workload's shape (different PRNG stream), no
             data-path collective: weak scaling, value = total Mbp of all ranks / max-over-ranks time.
  lcb      : ONE alignment (the same genomes on every rank); the independent units of the path -- the pairwise finder
             passes of the guide tree, the recursion's gaps, the gapped-DP intervals -- are LPT-sharded over the ranks and
             their results exchanged with RCCL all_gathers (mauvealigner_amd/parallel.py): strong scaling.

Extra objects on the JSON line:
  roofline     -- dominant HBM kernel (by HIP-event time on the library's stream): algorithmic bytes per launch
                   / average launch duration, against the 8 TB/s HBM peak; the seed-pass
                  aggregate (SURVEY.md 8d's B_seed per position) and the DP kernel's GCUPS beside it.
  cpu_baseline -- the CPU oracle (oracle/, a restatement: kind "port") timed on this box's host cores on the
                  same workload or a bounded sample of it (rank 0, N = 1 only).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# DP: 3-state Gotoh, ~12 integer ops per cell: CUs * 64 lanes * clock / 12
DP_PEAK_GCUPS = 256 * 64 * 2.4 / 12.0

WORKLOADS = {
    "C2": dict(n=3, L=5_000_000, weight=15, path="align", key=2,
               text="C2: 3 x %d bp synthetic genomes, ~3%% divergence, seed weight 15"),
    "C3": dict(n=5, L=5_000_000, weight=15, path="align", key=3,
               text="C3: 5 x %d bp synthetic genomes, ~3%% divergence, ~50 inversions, seed weight 15"),
    "C4": dict(n=8, L=2_000_000, weight=0, path="progressive", key=4,
               text="C4: 8 x %d bp synthetic genomes on a balanced tree (branch divergence 0.01, 2 inversions + 1 clade insertion per branch), progressive path, default seed weight"),
    "C5": dict(n=2, L=100_000_000, weight=0, path="align", key=5,
               text="C5: 2 x %d bp synthetic genomes, ~3%% divergence, 2000 lineage-specific insertions + 200 hyper-divergent segments per genome, default seed weight (19)"),
}


def algorithmic_bytes_per_position(weight):
    """B_seed = 0.25 + (K+4) + R*2*(K+4) + (K+4); also the per-kernel split."""
    K = 4 if 2 * weight <= 32 else 8
    R = (2 * weight + 7) // 8
    per_kernel = {
        "seed_extract": 0.25 + (K + 4),          # read 2-bit base, write key + pos
        "rs_scatter": 2.0 * (K + 4),             # one radix pass: read + write of key + pos
        "rs_hist": 0.0,                          # re-read of the keys: overhead, not algorithmic
        "rs_rowscan": 0.0,
        "mum_join": float(K + 4),                # read key + pos once
    }
    total = 0.25 + (K + 4) + R * 2 * (K + 4) + (K + 4)
    return total, per_kernel, K, R


def make_workload(name, scale, rank, track=False):
    """-> (genomes, origins or None, nominal genome length)"""
    from mauvealigner_amd import synth
    w = WORKLOADS[name]
    L = int(w["L"] * scale)
    if name in ("C2", "C3"):
        inv = 0 if name == "C2" else (50 if scale >= 0.25 else max(1, int(round(50 * min(1.0, scale * 4)))))
        out = synth.star_genomes(w["n"], L, 0.03, w["key"] + 1000 * rank, inversions=inv, track=track)
        return (out[0], out[1], L) if track else (out, None, L)
    if rank:
        raise SystemExit("bench.py: %s is a single-alignment workload (use --shard lcb with --gpus N)" % name)
    return synth.make_config(name, scale), None, L


def pack_pinned(genomes):
    """the packed genomes in host RAM, in page-locked caller buffers (where the timed region starts)"""
    from mauvealigner_amd import _lib
    ws = [_lib.pack_codes(g) for g in genomes]
    block = _lib.pinned_empty(sum(len(w) for w in ws), np.uint64)      # one block, genome behind genome: the upload is one DMA
    out, at = [], 0
    for w in ws:
        block[at:at + len(w)] = w
        out.append(block[at:at + len(w)])
        at += len(w)
    return out, [len(g) for g in genomes]


def seed_weight_of(cfg, genomes):
    from mauvealigner_amd import _lib
    return cfg["weight"] or _lib.default_seed_weight(sum(len(g) for g in genomes) // len(genomes))


class Runner:
    """one workload on one context: the host-to-host step and the device-resident pass"""

    def __init__(self, ctx, name, genomes, params, barrier):
        from mauvealigner_amd import _lib
        self.ctx, self.name, self.cfg, self.params, self.barrier = ctx, name, WORKLOADS[name], params, barrier
        self.genomes = genomes
        self.packed, self.lens = pack_pinned(genomes)
        self.bufs = _lib.ResultBuffers()
        self.total_bp = sum(self.lens)
        self.progressive = self.cfg["path"] == "progressive"

    def upload(self):
        self.ctx.set_genomes_packed(self.packed, self.lens)

    def run(self, params=None, fetch=False, out=None, compact=False):
        p = params or self.params
        if self.progressive:
            return self.ctx.progressive_align(p, fetch=fetch, out=out, compact=compact)
        return self.ctx.align(p, fetch=fetch, out=out, compact=compact)

    def step_host(self, params=None, compact=True):       # host RAM -> host RAM
        # every result array comes back, in the narrowest types that hold it (mauve_align_fetch_compact: a byte per column for up to 8 genomes, int32
        # tables); compact=False: the int64 / uint32 arrays of mauve_align_fetch (reported beside `value` as `fetch_wide`)
        self.upload()
        return self.run(params, fetch=True, out=self.bufs, compact=compact)

    def time_host(self, steps, warmup, params=None, compact=True):
        for _ in range(warmup):
            self.step_host(params, compact)
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = self.step_host(params, compact)
        self.barrier()
        return time.perf_counter() - t0, r

    def time_resident(self, steps, warmup, params=None):
        self.upload()
        for _ in range(warmup):
            self.run(params)
        acc = {}
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            sizes = self.run(params)
            for k, v in self.ctx.stage_times().items():
                acc[k] = acc.get(k, 0.0) + v
        self.barrier()
        sizes = {k: v for k, v in sizes.items() if isinstance(v, int)}
        return time.perf_counter() - t0, {k: round(v / steps, 3) for k, v in acc.items()}, sizes


def kernel_source_digest():
    h = hashlib.sha256()
    for f in ("seed_pass.hip", "dp_batch.hip", "common.hpp", "dev_scan.hpp"):
        with open(os.path.join(ROOT, "mauvealigner_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def committed_traffic(kernel, config="C3"):
    """roofline.traffic: HBM bytes per launch from the committed PMC passes (profiles/roofline_traffic.json, made by
    tools/summarize_profile.py from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this command).  The file carries the
    digest of the kernel sources it was measured on; when the sources have changed since, the figure is dropped (null)
    rather than quoted stale."""
    tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    try:
        with open(tpath) as f:
            t = json.load(f)
    except Exception:
        return None, "no committed PMC profile"
    if config != "C3":
        t = t.get("_configs", {}).get(config, {})
        if not t:
            return None, "no committed PMC profile of %s" % config
    meta = t.get("_meta", {})
    if meta.get("kernel_source_digest") != kernel_source_digest():
        return None, "committed PMC profile (%s) predates the current kernel sources" % meta.get("round", "?")
    v = t.get(kernel, {}).get("hbm_bytes_per_launch")
    return v, "profiles/roofline_traffic.json (%s, commit %s)" % (meta.get("round", "?"), meta.get("commit", "?"))


def kernel_profile(rn, weight, nprof=3, params=None):
    """per-kernel HIP-event timing on the library's stream (separate, untimed passes; events serialize the launches)"""
    ctx = rn.ctx
    rn.upload()
    rn.run(params)
    ctx.profile(True)
    ctx.profile_reset()
    for _ in range(nprof):
        sizes = rn.run(params)
    ctx.profile(False)
    kernels = ctx.profile_get()
    tot_b, per_kernel, K, R = algorithmic_bytes_per_position(weight)
    timed = [k for k in kernels if kernels[k]["launches"]]
    # the dominant kernel of a pass: the one kernel with the most time in it (canon_sort / misc_sort are classes of ~20 small launches, not kernels)
    overall = max((k for k in timed if k not in ("canon_sort", "misc_sort")), key=lambda k: kernels[k]["ms"])
    hbm_kernels = [k for k in timed if per_kernel.get(k)]
    dom = max(hbm_kernels, key=lambda k: kernels[k]["ms"])
    d = kernels[dom]
    avg_ms = d["ms"] / d["launches"]
    units = d["units"] / d["launches"]
    bpp = per_kernel[dom]
    achieved = bpp * units / (avg_ms * 1e-3) / 1e9
    traffic, tnote = committed_traffic(dom, rn.name)
    hbm = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
           "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": tnote,
           "algorithmic_bytes_per_launch": bpp * units, "avg_launch_ms": round(avg_ms, 4), "launches_timed": d["launches"]}
    dp_ms = kernels["dp_step"]["ms"] / nprof
    cells = sizes["n_dp_cells"]
    dp_traffic, dp_tnote = committed_traffic("dp_step2", rn.name)
    dp = {"bound": "valu", "kernel": "dp_step (dp_step2 + dp_step_wide side by side)", "cells": cells, "kernel_ms": round(dp_ms, 4),
          "achieved": round(cells / (dp_ms * 1e-3) / 1e9, 2) if dp_ms else None, "peak": round(DP_PEAK_GCUPS, 1), "unit": "GCUPS",
          "frac": round(cells / (dp_ms * 1e-3) / 1e9 / DP_PEAK_GCUPS, 4) if dp_ms else None,
          "peak_note": "256 CUs x 64 lanes x 2.4 GHz / 12 integer operations per cell (SURVEY.md 8d): no HBM or MFMA roofline applies to the gapped DP",
          "traffic": dp_traffic, "traffic_source": dp_tnote, "launches_timed": kernels["dp_step"]["launches"]}
    dp["gcups"] = dp["achieved"]; dp["peak_gcups"] = dp["peak"]
    if not rn.progressive and params is None:
        # the launch's critical path, measured: the DP of the single largest interval alone (mauve_align_begin / mauve_align_dp, same kernels).  The
        # progressive steps of one interval are serial and one wave runs them, so the launch cannot be shorter than this whatever the other intervals do.
        try:
            import numpy as np
            n_dp, cost, cap = ctx.align_begin(rn.params)
            if n_dp:
                top = np.array([int(np.argmax(cost))])
                ctx.profile(True)
                for _ in range(2):
                    ctx.profile_reset(); ctx.align_dp(top, cap)
                one = ctx.profile_get()["dp_step"]["ms"]
                ctx.profile(False)
                dp["largest_interval_alone"] = {"kernel_ms": round(one, 4), "share_of_launch": round(one / dp_ms, 3) if dp_ms else None, "cell_bound": int(cost[top[0]]),
                                                "intervals": int(n_dp), "note": "the DP launch is bounded below by its largest interval's serial lines (DESIGN.md section 9)"}
        except Exception as e:                                   # (a measurement beside the line, never a reason to lose the line)
            dp["largest_interval_alone"] = {"error": str(e)[:200]}
    # the line's roofline object describes the DOMINANT kernel of the pass, whatever bounds it; the dominant HBM kernel and the DP ride beside it
    roofline = dict(dp if overall == "dp_step" else (hbm if overall == dom else
                    {"bound": "hbm" if per_kernel.get(overall) else "latency (gather / look-back: no HBM or MFMA roofline applies)", "kernel": overall,
                     "achieved": round(per_kernel.get(overall, 0.0) * kernels[overall]["units"] / kernels[overall]["ms"] / 1e6, 1) if per_kernel.get(overall) else None,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(per_kernel.get(overall, 0.0) * kernels[overall]["units"] / kernels[overall]["ms"] / 1e6 / HBM_PEAK_GBS, 4) if per_kernel.get(overall) else None,
                     "traffic": committed_traffic(overall, rn.name)[0]}))
    roofline["dominant_by"] = "kernel time per pass (HIP events): %s %.4f ms" % (overall, kernels[overall]["ms"] / nprof)
    roofline["hbm_kernel"] = hbm
    roofline["dp"] = dp
    seed_k = ["seed_extract", "rs_hist", "rs_rowscan", "rs_scatter", "mum_join"]
    seed_ms = sum(kernels[k]["ms"] for k in seed_k) / nprof
    P = kernels["seed_extract"]["units"] / nprof                     # positions of all seed passes of one step
    ext_ms = (kernels["mum_runs"]["ms"] + kernels["mum_extend"]["ms"]) / nprof
    roofline["seed_pass"] = {"bytes_per_position": tot_b, "positions": P, "kernel_ms": round(seed_ms, 4),
                             "achieved_GBs": round(tot_b * P / (seed_ms * 1e-3) / 1e9, 1) if seed_ms else None,
                             "frac": round(tot_b * P / (seed_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if seed_ms else None,
                             "with_runs_and_extend": {"kernel_ms": round(seed_ms + ext_ms, 4),
                                                      "frac": round(tot_b * P / ((seed_ms + ext_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if seed_ms else None},
                             "seed_passes_per_step": kernels["seed_extract"]["launches"] // nprof,
                             "sort_passes_run": kernels["rs_scatter"]["launches"] // nprof,
                             "note": "B_seed is SURVEY 8(d)'s figure for a full LSD sort of ceil(2w/8) passes; the N-way seed pass "
                                     "sorts the high mer bits only and joins through an LDS hash table (DESIGN.md section 4)"}
    kern_ms = {k: round(v["ms"] / nprof, 4) for k, v in kernels.items()}
    return roofline, kern_ms


KEYS_ALIGN = ("anchor_start", "anchor_length", "left", "right", "reverse", "col_off", "cols", "dp_score")
KEYS_PROG = ("left", "right", "reverse", "col_off", "cols", "dp_score")


def cpu_baseline(name, scale, weight, gpu_result, genomes, sample_scale):
    """the oracle on the workload (sample_scale = 1.0: the very genomes, doubling as the full-size parity check of the GPU
    result) or on a bounded sample of the same shape (a smaller scale of the same generator)"""
    from oracle import pyoracle as O
    from mauvealigner_amd import synth
    cfg = WORKLOADS[name]
    if sample_scale >= 1.0:
        gs = genomes
    else:
        gs = synth.make_config(name, scale * sample_scale)
    kw = dict(seed_weight=weight) if cfg["weight"] else {}
    tc0 = time.perf_counter()
    if cfg["path"] == "progressive":                      # the progressiveMauve call site's option set, like the GPU leg (params_for)
        ref = O.progressive_align(gs, O.default_progressive_params(**kw))
    else:
        ref = O.align(gs, O.default_params(**kw))
    tc = time.perf_counter() - tc0
    bp = sum(len(g) for g in gs)
    cpu = {"value": round(bp / 1e6 / tc, 3), "unit": "Mbp/s", "cores": 1, "kind": "port",
           "sample": ("full workload" if sample_scale >= 1.0 else "the same generator at %.3g of the size" % sample_scale) +
                     " (%d x ~%d bp), single thread, one pass, %.1f s" % (len(gs), bp // len(gs), tc),
           "host_cpus": os.cpu_count()}
    if sample_scale >= 1.0 and gpu_result is not None:
        keys = KEYS_PROG if cfg["path"] == "progressive" else KEYS_ALIGN
        cpu["gpu_result_identical"] = bool(all(np.array_equal(gpu_result[k], ref["aln"][k]) for k in keys))
    return cpu


def all_cores_baseline(config, scale, weight, wscale_rel=0.2):
    """the same baseline on every core this process may use: independent copies of the workload (the oracle is single-threaded like the reference's
    default build, and the reference's optional OpenMP build parallelises over the same independent units -- LCB intervals -- so throughput of
    independent copies is its upper bound), at a fraction of the size so that the memory of all copies fits"""
    import subprocess
    try:
        ncore = max(1, len(os.sched_getaffinity(0)))
        wscale = wscale_rel * scale
        ws = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_worker", config, str(wscale), str(weight)], cwd=ROOT,
                               stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True) for _ in range(ncore)]
        for w in ws:
            assert w.stdout.readline().strip() == "ready"
        tw0 = time.perf_counter()
        for w in ws:
            w.stdin.write("go\n"); w.stdin.flush()
        bases = 0
        for w in ws:
            bases += int(w.stdout.readline().split()[1])
        tw = time.perf_counter() - tw0
        for w in ws:
            w.wait(timeout=60)
        return {"value": round(bases / 1e6 / tw, 2), "unit": "Mbp/s", "cores": ncore,
                "sample": "%d concurrent copies of the workload at scale %.2f, %.1f s" % (ncore, wscale, tw)}
    except Exception as ex:                      # the single-core figure stands on its own
        return {"error": repr(ex)}


def params_for(name, **kw):
    """the reference call site's defaults for the workload's path: mauveAligner.cpp:92-99 (mauve_default_params) for the mauveAligner
    configs, progressiveMauve.cpp:578-579,624-637 (mauve_default_progressive_params: SP scoring, weight scaling 0.5 / 0.5,
    refinement on) for the progressive one"""
    from mauvealigner_amd import _lib
    cfg = WORKLOADS[name]
    if cfg["weight"]:
        kw.setdefault("seed_weight", cfg["weight"])
    if cfg["path"] == "progressive":
        return _lib.default_progressive_params(**kw)
    return _lib.default_params(**kw)


def option_set(params, progressive):
    o = {"extend_lcbs": int(params.extend_lcbs), "max_extension_iters": int(params.max_extension_iters), "recursive": int(params.recursive),
         "gapped": int(params.gapped)}
    if progressive:
        o.update(lcb_scoring="sp" if params.lcb_scoring == 1 else "length", weight_scaling=int(params.weight_scaling),
                 conservation_scale_ppm=int(params.conservation_scale_ppm), bp_dist_scale_ppm=int(params.bp_dist_scale_ppm),
                 refine_rounds=int(params.refine_rounds), seed_family=int(params.seed_family),
                 source="mauve_default_progressive_params (progressiveMauve.cpp:578-579,624-637)" if params.lcb_scoring == 1 and params.refine_rounds == 2 and params.weight_scaling
                 else "mauve_default_params: last round's option set, NOT the call site's defaults")
    else:
        o["source"] = "mauve_default_params (mauveAligner.cpp:92-99)"
    return o


def secondary_leg(ctx, name, scale, steps, warmup, barrier, cpu_sample):
    """one of the other BASELINE configs, measured like the primary: host-to-host value, resident pass, stages, roofline, CPU"""
    cfg = WORKLOADS[name]
    tg0 = time.perf_counter()
    genomes, _, L = make_workload(name, scale, 0)
    tgen = time.perf_counter() - tg0
    params = params_for(name)
    rn = Runner(ctx, name, genomes, params, barrier)
    weight = seed_weight_of(cfg, genomes)
    e_h, res = rn.time_host(steps, warmup)
    gpu_result = {k: np.array(v, copy=True) for k, v in res.items() if isinstance(v, np.ndarray)} if cpu_sample >= 1.0 else None
    e_r, stages, sizes = rn.time_resident(steps, 1)
    e_w, _ = rn.time_host(max(2, steps // 2), 1, compact=False)
    roof, kern = kernel_profile(rn, weight, nprof=2)
    out = {"workload": cfg["text"] % L, "metric": "aligned Mbp/s (seed+extend+DP)", "value": round(rn.total_bp / 1e6 / (e_h / steps), 2), "unit": "Mbp/s",
           "ms_per_step": round(e_h / steps * 1e3, 3), "steps": steps,
           "timed_region": "packed genomes in page-locked host RAM -> upload -> %s -> every result array in page-locked host RAM (mauve_align_fetch_compact)" %
                           ("mauve_progressive_align" if rn.progressive else "mauve_align"),
           "fetch_wide": {"ms_per_step": round(e_w / max(2, steps // 2) * 1e3, 3), "note": "the same step with mauve_align_fetch (uint32 columns, int64 tables)"},
           "total_bp": rn.total_bp, "seed_weight": weight, "extend_lcbs": int(params.extend_lcbs),
           "config": option_set(params, rn.progressive),
           "device_resident": {"ms_per_step": round(e_r / steps * 1e3, 3), "Mbp_s": round(rn.total_bp / 1e6 / (e_r / steps), 2),
                               "note": "genomes resident in HBM, results left there"},
           "stages_ms": stages, "kernels_ms": kern, "roofline": roof, "result_sizes": sizes, "generate_s": round(tgen, 1)}
    if cpu_sample:
        out["cpu_baseline"] = cpu_baseline(name, scale, weight, gpu_result, genomes, cpu_sample)
        if name in ("C4", "C5"):           # every host core: independent copies of the workload at a size whose copies fit the memory together
            out["cpu_baseline"]["all_cores"] = all_cores_baseline(name, scale, weight if cfg["weight"] else 0, 0.1 if name == "C4" else 0.01)
    if rn.progressive:
        # last round's option set (length-weighted LCBs, no weight scaling, no refinement) as a named extra, never the leg's value
        from mauvealigner_amd import _lib
        pl = _lib.default_params(**({"seed_weight": cfg["weight"]} if cfg["weight"] else {}))
        e_l, _ = rn.time_host(steps, 1, pl)
        out["length_scoring_no_refinement"] = {"value": round(rn.total_bp / 1e6 / (e_l / steps), 2), "unit": "Mbp/s",
                                               "ms_per_step": round(e_l / steps * 1e3, 3), "config": option_set(pl, True)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--shard", default="replicas", choices=["replicas", "lcb"])
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the genomes (debug only; 1.0 = BASELINE size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", "--only-primary", dest="no_secondary", action="store_true",
                    help="skip the side measurements (the other configs, the pass with the other lcb_extension setting, the backbone call)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        print("bench.py: --gpus %d needs torch.distributed.run with WORLD_SIZE=%d" % (args.gpus, args.gpus), file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the hot path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("MAUVE_BENCH_FORCE_DIST"):      # the env hook lets a 1-GPU box exercise the RCCL path
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")           # (a forced single-rank rehearsal outside torchrun has no rendezvous in its environment)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    from mauvealigner_amd import _lib

    cfg = WORKLOADS[args.config]
    shard_lcb = args.shard == "lcb" and dist is not None
    if cfg["path"] == "progressive" or args.config == "C5":
        if dist is not None and not shard_lcb and world > 1:
            print("bench.py: %s with --gpus N needs --shard lcb" % args.config, file=sys.stderr)
            sys.exit(2)
    genomes, origins, L = make_workload(args.config, args.scale, 0 if shard_lcb else rank, track=args.config in ("C2", "C3"))
    total_bp = sum(len(g) for g in genomes)
    ctx = _lib.Context(local_rank)
    weight = seed_weight_of(cfg, genomes)
    params = params_for(args.config)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    rn = Runner(ctx, args.config, genomes, params, barrier)
    t_up0 = time.perf_counter()
    rn.upload()                                   # first upload: allocates the device buffers
    t_upload = time.perf_counter() - t_up0

    rccl_comm = None
    if shard_lcb:
        from mauvealigner_amd import parallel
        # mauve_set_shard_rccl: the same calls on every rank, the independent units dealt out inside them, the exchanges run by the library itself
        # (ncclAllGather on its stream); torch.distributed only carries rank 0's unique id to the others.  MAUVE_BENCH_SHARD_CALLBACK: the callback
        # form through torch.distributed (mauve_set_shard) instead.
        if os.environ.get("MAUVE_BENCH_SHARD_CALLBACK"):
            parallel.attach_shard(ctx, dist)
        else:
            box = [parallel.RcclComm.new_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            rccl_comm = parallel.RcclComm(rank, world, box[0])
            parallel.attach_shard_rccl(ctx, rccl_comm)

        def step():
            return rn.step_host()
        for _ in range(args.warmup):
            res = step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = step()
        barrier()
        elapsed = time.perf_counter() - t0
    else:
        elapsed, res = rn.time_host(args.steps, args.warmup)
    sizes = {k: v for k, v in res.items() if isinstance(v, int)}
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tb = torch.tensor([float(total_bp)], dtype=torch.float64, device="cuda")
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        total_all = float(total_bp) if shard_lcb else float(tb.item())
    else:
        total_all = float(total_bp)
    ms_per_step = elapsed / args.steps * 1e3
    value = total_all / 1e6 / (elapsed / args.steps)
    side = rank == 0 and world == 1                          # the side measurements: one GPU, rank 0
    gpu_result = {k: np.array(v, copy=True) for k, v in res.items() if isinstance(v, np.ndarray)} if side else None

    roofline, kern_ms, extras, stages = None, {}, {}, {}
    if rank == 0 and not shard_lcb:
        nw = max(3, args.steps // 2)
        e_w, _ = rn.time_host(nw, 1, compact=False)
        extras["fetch_wide"] = {"ms_per_step": round(e_w / nw * 1e3, 3), "Mbp_s": round(total_bp / 1e6 / (e_w / nw), 2),
                                "note": "the same step with mauve_align_fetch (uint32 columns, int64 tables) instead of mauve_align_fetch_compact"}
        # ---- the same pass with the genomes resident in HBM and the results left there (a named extra, never `value`) ----
        e_r, stages, _ = rn.time_resident(args.steps, 1)
        extras["device_resident"] = {"ms_per_step": round(e_r / args.steps * 1e3, 3), "Mbp_s": round(total_bp / 1e6 / (e_r / args.steps), 2),
                                     "note": "genomes resident in HBM, results left there (mauve_align only)"}
        roofline, kern_ms = kernel_profile(rn, weight)
        extras["kernel_share_of_resident_pass"] = round(sum(kern_ms.values()) / (e_r / args.steps * 1e3), 3)

    # ---- the other setting of lcb_extension (mauveAligner.cpp:95: the reference's default is on) ----
    if side and not args.no_secondary and not rn.progressive:
        n2 = max(3, args.steps // 2)
        other = 0 if params.extend_lcbs else 1
        pe = params_for(args.config, extend_lcbs=other)
        e4, r4 = rn.time_host(n2, 1, pe)
        extras["lcb_extension_%s" % ("on" if other else "off")] = {
            "ms_per_step": round(e4 / n2 * 1e3, 3), "Mbp_s": round(total_bp / 1e6 / (e4 / n2), 1), "timed_region": "host RAM -> host RAM, as `value`",
            "n_anchor": r4["n_anchor"], "n_lcb": r4["n_lcb"]}
        if origins is not None:
            from mauvealigner_amd import accuracy
            acc_e = accuracy.score_alignment(r4, origins)
            extras["lcb_extension_%s" % ("on" if other else "off")].update(sensitivity=round(acc_e["sensitivity"], 5), ppv=round(acc_e["ppv"], 5))
        if not other and roofline is not None:
            # the seed-pass figure of the MAIN pass alone: with the extension on, a step's kernel time also holds the small extension passes
            # (a few thousand windows each: launches, no positions worth counting), which B_seed x P does not count
            roof_off, kern_off = kernel_profile(rn, weight, nprof=2, params=pe)
            roofline["seed_pass"]["main_pass_only"] = {k: roof_off["seed_pass"][k] for k in ("kernel_ms", "achieved_GBs", "frac", "with_runs_and_extend", "seed_passes_per_step")}
            roofline["seed_pass"]["main_pass_only"]["how"] = "the same step with extend_lcbs = 0 (one seed pass per step)"

    # ---- accuracy of the bench workload's alignment against the generator's truth (not timed) ----
    acc = None
    if side and origins is not None:
        from mauvealigner_amd import accuracy
        acc = accuracy.score_alignment(gpu_result, origins)
        acc = {k: (round(v, 5) if isinstance(v, float) else v) for k, v in acc.items()}

    # ---- the backbone stage on the columns the pass left in HBM: reported beside `value`, never part of it ----
    if side and not args.no_secondary and not rn.progressive:
        n2 = max(3, args.steps // 2)
        rn.upload()
        rn.run()
        ctx.backbone(island_gap=20)
        t0 = time.perf_counter()
        for _ in range(n2):
            bb = ctx.backbone(island_gap=20)
        eb = time.perf_counter() - t0
        extras["backbone"] = {"ms_per_call": round(eb / n2 * 1e3, 3), "segments": int(len(bb["seg_iv"])), "islands": int(len(bb["islands"])),
                              "island_gap": 20}
        # the homology pass in front of it (detectAndApplyBackbone's HMM as a two-state Viterbi scan), on the same resident result
        th = 0.0
        for _ in range(n2):
            rn.run()
            t0 = time.perf_counter()
            hres = ctx.apply_homology(fetch=False)
            th += time.perf_counter() - t0
        extras["homology_pass"] = {"ms_per_call": round(th / n2 * 1e3, 3), "residues_moved": int(hres["n_moved"]), "columns_after": int(hres["n_cols"]),
                                   "hmm": "identity 0.7, pgh 1e-5, pgu 1e-9 (progressiveMauve.cpp:319-322)"}

    # ---- CPU baseline: the oracle on the same workload, host cores of this box ----
    cpu = None
    if side and not args.no_cpu_baseline:
        sample = 1.0 if args.config in ("C2", "C3", "C4") else 0.1
        cpu = cpu_baseline(args.config, args.scale, weight, gpu_result, genomes, sample)
        cpu["all_cores"] = all_cores_baseline(args.config, args.scale, weight if cfg["weight"] else 0, 0.2 if args.config != "C5" else 0.01)

    # ---- the other BASELINE configs ----
    legs = {}
    if side and not args.no_secondary:
        del rn
        plan = [("C2", max(3, args.steps // 2), 0), ("C4", max(3, args.steps // 4), 1.0), ("C5", max(3, args.steps // 4), 0.1)]
        for name, st, cpu_sample in plan:
            if name == args.config:
                continue
            try:
                legs[name.lower()] = secondary_leg(ctx, name, args.scale, st, 1, barrier, 0 if args.no_cpu_baseline else cpu_sample)
            except Exception as ex:                # a side leg never takes the headline down with it
                legs[name.lower()] = {"error": repr(ex)}

    if rank == 0:
        if shard_lcb:
            from mauvealigner_amd import parallel
            st = parallel.shard_stats(ctx)
            nst = args.steps + args.warmup
            extras["shard_exchanges"] = {"per_step": round(st["exchanges"] / nst, 2), "bytes_sent_per_step": int(st["bytes_sent"] / nst),
                                         "bytes_received_per_step": int(st["bytes_received"] / nst), "ms_per_step": round(st["ms"] / nst, 3),
                                         "collective": "callback (torch.distributed)" if rccl_comm is None else "ncclAllGather inside the library (mauve_set_shard_rccl)"}
        par = ("one alignment: pairwise finder passes, recursion gaps and DP intervals LPT-sharded over the ranks, results exchanged with RCCL all_gathers" if shard_lcb
               else "independent genome sets per GPU (no data-path collective)")
        out = {
            "metric": "aligned Mbp/s (seed+extend+DP)", "value": round(value, 2), "unit": "Mbp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong" if shard_lcb else "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": (cfg["text"] % L) + ", recursive anchoring + gapped DP on",
                       "genomes_per_gpu": cfg["n"], "genome_length": L, "seed_weight": weight, "parallelism": par,
                       "extend_lcbs": int(params.extend_lcbs), "max_extension_iters": int(params.max_extension_iters),
                       "options": option_set(params, cfg["path"] == "progressive"),
                       "timed_region": "packed genomes in page-locked host RAM -> upload -> %s -> every result array in page-locked host RAM, narrowest types (mauve_align_fetch_compact; SURVEY.md 8d)" %
                                       ("mauve_progressive_align" if cfg["path"] == "progressive" else "mauve_align")},
            "roofline": roofline, "cpu_baseline": cpu, "accuracy_vs_truth": acc,
            "stages_ms": stages, "kernels_ms": kern_ms, "result_sizes": sizes, "upload_ms": round(t_upload * 1e3, 2), "device": ctx.device_name(),
            "prng": "xoshiro256** seeded through splitmix64, base seed 0x4D41555645 + config id (SURVEY 8d; mauvealigner_amd/synth.py, csrc/synth_rng.c)",
        }
        out.update(extras)
        out.update(legs)
        print(json.dumps(out, default=lambda o: o.tolist() if hasattr(o, 'tolist') else repr(o)))
    ctx.close()
    if rccl_comm is not None:
        rccl_comm.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
