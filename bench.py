#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native mauveAligner hot path.

Metric (BASELINE.json): aligned Mbp/s (seed + extend + chain + recursive anchoring + gapped DP) on N x 5 Mbp
genomes.  A "step" is one pass of the whole hot path (mauve_align) over one synthetic genome set whose packed
genomes are already resident in HBM; the timed region ends with the SoA results in host RAM.

Workload at every rank: BASELINE config C2 -- 3 x 5 Mbp E.-coli-scale synthetic genomes, ~3 % divergence, seed
weight 15 (configs[1], the configuration the metric is quoted on for one GPU).  With --gpus N every rank aligns
its own C2-shaped genome set (different PRNG stream), no data-path collective: weak scaling, value = total Mbp
of all ranks / max-over-ranks time.

Extra objects on the JSON line:
  roofline     -- dominant kernel (by HIP-event time on the library's stream), algorithmic bytes per launch
                  (DESIGN.md "Roofline accounting") / average launch duration, against the 8 TB/s HBM peak.
  cpu_baseline -- the CPU oracle (oracle/, a restatement: kind "port") timed on this box's host cores on the
                  same workload (rank 0, N=1 only).
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the genomes (debug only; 1.0 = BASELINE size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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

    from mauvealigner_amd import _lib, synth

    # ---- workload: C2-shaped genome set, one per rank ----
    weight = 15
    L = int(5_000_000 * args.scale)
    genomes, origins = synth.star_genomes(3, L, 0.03, 2 + 1000 * rank, track=True)   # origins: truth, for accuracy only
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

    for _ in range(args.warmup):
        sizes = ctx.align(params, fetch=False)
    barrier()
    t0 = time.perf_counter()
    stage_acc = {}
    for _ in range(args.steps):
        sizes = ctx.align(params, fetch=False)
        for k, v in ctx.stage_times().items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tb = torch.tensor([float(total_bp)], dtype=torch.float64, device="cuda")
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        total_all = float(tb.item())
    else:
        total_all = float(total_bp)
    ms_per_step = elapsed / args.steps * 1e3
    value = total_all / 1e6 / (elapsed / args.steps)

    # ---- per-kernel HIP-event timing (separate, untimed passes; events serialize the launches) ----
    roofline = None
    kernels = {}
    if rank == 0:
        ctx.profile(True)
        ctx.profile_reset()
        nprof = 3
        for _ in range(nprof):
            ctx.align(params, fetch=False)
        ctx.profile(False)
        kernels = ctx.profile_get()
        tot_b, per_kernel, K, R = algorithmic_bytes_per_position(weight)
        # The roofline object is for the dominant HBM-streaming kernel.  dp_step and mum_extend are VALU / shuffle /
        # L2-gather bound: neither an HBM nor an MFMA roofline applies to them, so they are
        # listed with their times beside it instead of being priced against the wrong peak.
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
        P = kernels["mum_join"]["units"] / max(1, kernels["mum_join"]["launches"])
        roofline["seed_pass"] = {"bytes_per_position": tot_b, "positions": P, "kernel_ms": round(seed_ms, 4),
                                 "achieved_GBs": round(tot_b * P / (seed_ms * 1e-3) / 1e9, 1) if seed_ms else None,
                                 "frac": round(tot_b * P / (seed_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if seed_ms else None}

    # ---- accuracy of the bench workload's alignment against the generator's truth (not timed) ----
    acc = None
    gpu_result = None
    if rank == 0 and world == 1:
        from mauvealigner_amd import accuracy
        gpu_result = ctx.align(params)
        acc = accuracy.score_alignment(gpu_result, origins)
        acc = {k: (round(v, 5) if isinstance(v, float) else v) for k, v in acc.items()}

    # ---- CPU baseline: the oracle on the same workload, host cores of this box ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle as O
        sample_scale = 1.0
        gs_cpu = genomes
        tc0 = time.perf_counter()
        ref = O.align(gs_cpu, O.default_params(seed_weight=weight))
        tc = time.perf_counter() - tc0
        # the same pass doubles as the full-size parity check of the GPU result (the oracle as the checker)
        import numpy as np
        parity = all(np.array_equal(gpu_result[k], ref["aln"][k]) for k in
                     ("anchor_start", "anchor_length", "left", "right", "reverse", "col_off", "cols", "dp_score"))
        cpu = {"value": round(sum(len(g) for g in gs_cpu) / 1e6 / tc, 3), "unit": "Mbp/s", "cores": 1, "kind": "port",
               "sample": "full workload (3 x %d bp, scale %.2f), single thread, one pass, %.1f s" % (L, sample_scale, tc),
               "host_cpus": os.cpu_count(), "gpu_result_identical": bool(parity)}
        # the same baseline on all the cores this process may use: independent copies of the workload, one per
        # worker, started together (the path has no intra-job CPU parallelism to offer; throughput adds up)
        try:
            import subprocess
            ncore = max(1, min(16, len(os.sched_getaffinity(0))))
            ws = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_worker", "C2", "1.0", str(weight)], cwd=ROOT,
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
                w.wait(timeout=30)
            cpu["all_cores"] = {"value": round(bases / 1e6 / tw, 2), "unit": "Mbp/s", "cores": ncore,
                                "sample": "%d concurrent copies of the workload, %.1f s" % (ncore, tw)}
        except Exception as ex:                      # the single-core figure above stands on its own
            cpu["all_cores"] = {"error": repr(ex)}

    if rank == 0:
        out = {
            "metric": "aligned Mbp/s (seed+extend+DP)", "value": round(value, 2), "unit": "Mbp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "C2: 3 x %d bp synthetic genomes per GPU, ~3%% divergence, seed weight 15, "
                                   "recursive anchoring + gapped DP on" % L,
                       "genomes_per_gpu": 3, "genome_length": L, "seed_weight": weight,
                       "parallelism": "independent genome sets per GPU (no data-path collective)"},
            "roofline": roofline, "cpu_baseline": cpu, "accuracy_vs_truth": acc,
            "stages_ms": {k: round(v / args.steps, 3) for k, v in stage_acc.items()},
            "kernels_ms": {k: round(v["ms"] / 3, 4) for k, v in kernels.items()},
            "result_sizes": sizes, "upload_ms": round(t_upload * 1e3, 2), "device": ctx.device_name(),
        }
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
