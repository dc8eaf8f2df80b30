#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native mauveAligner hot path.

Metric (BASELINE.json): aligned Mbp/s (seed + extend + chain + recursive anchoring + gapped DP) on N x 5 Mbp
genomes.  A "step" is one pass of the whole hot path (mauve_align) over one synthetic genome set whose packed
genomes are already resident in HBM; the timed region ends with the SoA results in host RAM.

Workload at N = 1: BASELINE config C3 -- 5 x 5 Mbp genomes, ~3 % divergence, ~50 inversions, seed weight 15: the
configuration the >= 50 Mbp/s target of `north_star` is quoted on.  C2 (3 x 5 Mbp, configs[1]) is measured beside
it and reported in the "c2" object.  --config picks another primary workload.

--gpus N, two forms (--shard):
  replicas (default): every rank aligns its own genome set of the workload's shape (different PRNG stream), no
             data-path collective: weak scaling, value = total Mbp of all ranks / max-over-ranks time.
  lcb      : ONE alignment (the same genomes on every rank); every rank runs the deterministic front (seed pass,
             chaining, recursion), the gapped-DP intervals are LPT-sharded over the ranks and the packed columns
             exchanged with one RCCL all_gather (mauvealigner_amd/parallel.py): strong scaling.

Extra objects on the JSON line:
  roofline     -- dominant HBM kernel (by HIP-event time on the library's stream): algorithmic bytes per launch
                   / average launch duration, against the 8 TB/s HBM peak; the seed-pass
                  aggregate (SURVEY.md 8d's B_seed per position) and the DP kernel's GCUPS beside it.
  cpu_baseline -- the CPU oracle (oracle/, a restatement: kind "port") timed on this box's host cores on the
                  same workload (rank 0, N = 1 only).
"""
import argparse
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
    "C2": dict(n=3, L=5_000_000, inversions=0, weight=15,
               text="C2: 3 x %d bp synthetic genomes, ~3%% divergence, seed weight 15"),
    "C3": dict(n=5, L=5_000_000, inversions=50, weight=15,
               text="C3: 5 x %d bp synthetic genomes, ~3%% divergence, ~50 inversions, seed weight 15"),
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


def make_workload(name, scale, rank):
    from mauvealigner_amd import synth
    w = WORKLOADS[name]
    L = int(w["L"] * scale)
    key = {"C2": 2, "C3": 3}[name] + 1000 * rank
    return synth.star_genomes(w["n"], L, 0.03, key, inversions=w["inversions"], track=True), L


def time_steps(ctx, params, steps, barrier, fetch=False):
    barrier()
    t0 = time.perf_counter()
    acc = {}
    sizes = None
    for _ in range(steps):
        sizes = ctx.align(params, fetch=fetch)
        for k, v in ctx.stage_times().items():
            acc[k] = acc.get(k, 0.0) + v
    barrier()
    return time.perf_counter() - t0, acc, sizes


def kernel_profile(ctx, params, weight, nprof=3):
    """per-kernel HIP-event timing (separate, untimed passes; events serialize the launches)"""
    ctx.profile(True)
    ctx.profile_reset()
    for _ in range(nprof):
        sizes = ctx.align(params, fetch=False)
    ctx.profile(False)
    kernels = ctx.profile_get()
    tot_b, per_kernel, K, R = algorithmic_bytes_per_position(weight)
    timed = [k for k in kernels if kernels[k]["launches"]]
    overall = max(timed, key=lambda k: kernels[k]["ms"])
    hbm_kernels = [k for k in timed if per_kernel.get(k)]
    dom = max(hbm_kernels, key=lambda k: kernels[k]["ms"])
    d = kernels[dom]
    avg_ms = d["ms"] / d["launches"]
    units = d["units"] / d["launches"]
    bpp = per_kernel[dom]
    achieved = bpp * units / (avg_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if os.path.exists(tpath):
        try:
            with open(tpath) as f:
                traffic = json.load(f).get(dom, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": bpp * units, "avg_launch_ms": round(avg_ms, 4),
                "launches_timed": d["launches"],
                "dominant_overall": {"kernel": overall, "ms_per_pass": round(kernels[overall]["ms"] / nprof, 4),
                                     "bound": "hbm" if per_kernel.get(overall) else "valu/shuffle (no HBM or MFMA roofline applies)"}}
    seed_k = ["seed_extract", "rs_hist", "rs_rowscan", "rs_scatter", "mum_join"]
    seed_ms = sum(kernels[k]["ms"] for k in seed_k) / nprof
    P = kernels["seed_extract"]["units"] / max(1, kernels["seed_extract"]["launches"])
    roofline["seed_pass"] = {"bytes_per_position": tot_b, "positions": P, "kernel_ms": round(seed_ms, 4),
                             "achieved_GBs": round(tot_b * P / (seed_ms * 1e-3) / 1e9, 1) if seed_ms else None,
                             "frac": round(tot_b * P / (seed_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if seed_ms else None,
                             "sort_passes_run": kernels["rs_scatter"]["launches"] // nprof,
                             "note": "B_seed is SURVEY 8(d)'s figure for a full LSD sort of ceil(2w/8) passes; the seed pass "
                                     "sorts the high mer bits only and joins through an LDS hash table (DESIGN.md section 4)"}
    dp_ms = kernels["dp_step"]["ms"] / nprof
    cells = sizes["n_dp_cells"]
    roofline["dp"] = {"bound": "valu", "cells": cells, "kernel_ms": round(dp_ms, 4),
                      "gcups": round(cells / (dp_ms * 1e-3) / 1e9, 2) if dp_ms else None,
                      "peak_gcups": round(DP_PEAK_GCUPS, 1),
                      "frac": round(cells / (dp_ms * 1e-3) / 1e9 / DP_PEAK_GCUPS, 4) if dp_ms else None}
    kern_ms = {k: round(v["ms"] / nprof, 4) for k, v in kernels.items()}
    return roofline, kern_ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--shard", default="replicas", choices=["replicas", "lcb"])
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the genomes (debug only; 1.0 = BASELINE size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the side measurements (C2, the pass with LCB extension, the backbone call)")
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
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    from mauvealigner_amd import _lib

    cfg = WORKLOADS[args.config]
    weight = cfg["weight"]
    shard_lcb = args.shard == "lcb" and dist is not None
    (genomes, origins), L = make_workload(args.config, args.scale, 0 if shard_lcb else rank)
    total_bp = sum(len(g) for g in genomes)
    ctx = _lib.Context(local_rank)
    t_up0 = time.perf_counter()
    ctx.set_genomes(genomes)            # H2D upload: outside the timed region (inputs resident in HBM)
    t_upload = time.perf_counter() - t_up0
    params = _lib.default_params(seed_weight=weight)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    if shard_lcb:
        from mauvealigner_amd import parallel

        def step():
            return parallel.align_sharded(ctx, params, dist, fetch=False)
        for _ in range(args.warmup):
            sizes = step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sizes = step()
        barrier()
        elapsed, stage_acc = time.perf_counter() - t0, {}
    else:
        for _ in range(args.warmup):
            sizes = ctx.align(params, fetch=False)
        elapsed, stage_acc, sizes = time_steps(ctx, params, args.steps, barrier)
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

    roofline, kern_ms, extras = None, {}, {}
    if rank == 0:
        roofline, kern_ms = kernel_profile(ctx, params, weight)
        ksum = sum(kern_ms.values())
        extras["kernel_share_of_step"] = round(ksum / ms_per_step, 3) if not shard_lcb else None

    # ---- SURVEY 8(d) variants of the figure (not `value`): results fetched into caller buffers; genomes uploaded
    # inside the timed region (host buffers at the boundary, PCIe-inclusive) ----
    if rank == 0 and world == 1:
        n2 = max(3, args.steps // 2)
        e2, _, _ = time_steps(ctx, params, n2, barrier, fetch=True)
        extras["with_fetch"] = {"ms_per_step": round(e2 / n2 * 1e3, 3), "Mbp_s": round(total_bp / 1e6 / (e2 / n2), 1)}
        packed = [_lib.pack_codes(g) for g in genomes]           # the packed genomes start in host RAM
        glens = [len(g) for g in genomes]
        ctx.set_genomes_packed(packed, glens)
        barrier()
        t0 = time.perf_counter()
        for _ in range(n2):
            ctx.set_genomes_packed(packed, glens)
            ctx.align(params, fetch=True)
        barrier()
        e3 = time.perf_counter() - t0
        extras["h2d_inclusive"] = {"ms_per_step": round(e3 / n2 * 1e3, 3), "Mbp_s": round(total_bp / 1e6 / (e3 / n2), 1),
                                   "note": "upload of the packed genomes (host RAM -> HBM), the pass, and the copy of all results into caller buffers"}

    # ---- accuracy of the bench workload's alignment against the generator's truth (not timed) ----
    acc = None
    gpu_result = None
    if rank == 0 and world == 1:
        from mauvealigner_amd import accuracy
        gpu_result = ctx.align(params)
        acc = accuracy.score_alignment(gpu_result, origins)
        acc = {k: (round(v, 5) if isinstance(v, float) else v) for k, v in acc.items()}

    # ---- the same workload with the reference's default lcb_extension on (mauveAligner.cpp:95), and the
    # backbone stage on the columns the pass left in HBM: reported beside `value`, never part of it ----
    if rank == 0 and world == 1 and not args.no_secondary:
        from mauvealigner_amd import accuracy
        n2 = max(3, args.steps // 2)
        pe = _lib.default_params(seed_weight=weight, extend_lcbs=1)
        ctx.align(pe, fetch=False)
        e4, _, _ = time_steps(ctx, pe, n2, barrier)
        acc_e = accuracy.score_alignment(ctx.align(pe), origins)
        extras["with_lcb_extension"] = {"ms_per_step": round(e4 / n2 * 1e3, 3), "Mbp_s": round(total_bp / 1e6 / (e4 / n2), 1),
                                        "sensitivity": round(acc_e["sensitivity"], 5), "ppv": round(acc_e["ppv"], 5)}
        ctx.align(params, fetch=False)
        ctx.backbone(island_gap=20)
        t0 = time.perf_counter()
        for _ in range(n2):
            bb = ctx.backbone(island_gap=20)
        eb = time.perf_counter() - t0
        extras["backbone"] = {"ms_per_call": round(eb / n2 * 1e3, 3), "segments": int(len(bb["seg_iv"])), "islands": int(len(bb["islands"])),
                              "island_gap": 20}

    # ---- CPU baseline: the oracle on the same workload, host cores of this box ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle as O
        tc0 = time.perf_counter()
        ref = O.align(genomes, O.default_params(seed_weight=weight))
        tc = time.perf_counter() - tc0
        # the same pass doubles as the full-size parity check of the GPU result (the oracle as the checker)
        parity = all(np.array_equal(gpu_result[k], ref["aln"][k]) for k in
                     ("anchor_start", "anchor_length", "left", "right", "reverse", "col_off", "cols", "dp_score"))
        cpu = {"value": round(total_bp / 1e6 / tc, 3), "unit": "Mbp/s", "cores": 1, "kind": "port",
               "sample": "full workload (%d x %d bp), single thread, one pass, %.1f s" % (cfg["n"], L, tc),
               "host_cpus": os.cpu_count(), "gpu_result_identical": bool(parity)}
        # the same baseline on every core this process may use: independent copies of the workload (the path has no
        # intra-job CPU parallelism to offer; throughput adds up), at 1/5 size so that the memory of all copies fits
        try:
            import subprocess
            ncore = max(1, len(os.sched_getaffinity(0)))
            wscale = 0.2 * args.scale
            ws = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_worker", args.config, str(wscale), str(weight)], cwd=ROOT,
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
            cpu["all_cores"] = {"value": round(bases / 1e6 / tw, 2), "unit": "Mbp/s", "cores": ncore,
                                "sample": "%d concurrent copies of the workload at scale %.2f, %.1f s" % (ncore, wscale, tw)}
        except Exception as ex:                      # the single-core figure above stands on its own
            cpu["all_cores"] = {"error": repr(ex)}

    # ---- C2 (configs[1]) beside the primary workload ----
    c2 = None
    if rank == 0 and world == 1 and args.config != "C2" and not args.no_secondary:
        (g2, _), L2 = make_workload("C2", args.scale, 0)
        ctx.set_genomes(g2)
        p2 = _lib.default_params(seed_weight=WORKLOADS["C2"]["weight"])
        for _ in range(max(1, args.warmup)):
            ctx.align(p2, fetch=False)
        e, st, sz = time_steps(ctx, p2, args.steps, barrier)
        r2, k2 = kernel_profile(ctx, p2, WORKLOADS["C2"]["weight"])
        bp2 = sum(len(g) for g in g2)
        c2 = {"workload": WORKLOADS["C2"]["text"] % L2, "value": round(bp2 / 1e6 / (e / args.steps), 2), "unit": "Mbp/s",
              "ms_per_step": round(e / args.steps * 1e3, 3), "stages_ms": {k: round(v / args.steps, 3) for k, v in st.items()},
              "kernels_ms": k2, "roofline": r2, "result_sizes": sz}

    if rank == 0:
        par = ("one alignment, DP intervals LPT-sharded over the ranks, columns exchanged with one RCCL all_gather" if shard_lcb
               else "independent genome sets per GPU (no data-path collective)")
        out = {
            "metric": "aligned Mbp/s (seed+extend+DP)", "value": round(value, 2), "unit": "Mbp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong" if shard_lcb else "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": (cfg["text"] % L) + ", recursive anchoring + gapped DP on",
                       "genomes_per_gpu": cfg["n"], "genome_length": L, "seed_weight": weight, "parallelism": par},
            "roofline": roofline, "cpu_baseline": cpu, "accuracy_vs_truth": acc,
            "stages_ms": {k: round(v / args.steps, 3) for k, v in stage_acc.items()},
            "kernels_ms": kern_ms, "result_sizes": sizes, "upload_ms": round(t_upload * 1e3, 2), "device": ctx.device_name(),
            "prng": "numpy PCG64 (SURVEY 8d names xoshiro256**; the workloads are defined by mauvealigner_amd/synth.py)",
        }
        out.update(extras)
        if c2:
            out["c2"] = c2
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
