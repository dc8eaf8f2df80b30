#!/usr/bin/env python3
"""Digest gpurun_out/prof_<tag> (tools/profile_gpu.sh) into profiles/: the rocprofv3 kernel-stats table, per-kernel
HBM traffic and LDS counters from the PMC passes, and profiles/roofline_traffic.json (read by bench.py for
roofline.traffic).

PMC handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE come from
separate passes, are in KiB, and on gfx950 FETCH_SIZE under-reports wide coalesced streaming reads by 2x.  Our
loads are 4-16 B per lane, so the factor is calibrated on a kernel with a known byte count in the same access
pattern: the main sort's rs_hist launch reads exactly 4*P key bytes and writes a negligible histogram.

The radix-sort kernels run in two very different sizes in one pass of the hot path: the main sort of the P seed keys
and ~20 small sorts (chaining, canonical order, DP launch list) of ~50 k entries that are launch-latency bound.  The
two are kept apart here (by grid size; the small ones are tagged [small]) -- one average over both describes neither.
Usage: python tools/summarize_profile.py <round-tag> [C4|C5|C2]   (e.g. r04, r04 C5).  Without a config: the C3 command, written to
profiles/<tag>_* and to the top level of profiles/roofline_traffic.json; with one: profiles/<tag>_<cfg>_* and the "_configs" section of that file.
"""
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = os.path.join(ROOT, "profiles")


def short(name):
    """Kernel name without arguments.  Sort kernels keep their key type, tiled compactions their functor."""
    name = name.replace("(anonymous namespace)::", "").replace("devscan::", "")
    m = re.match(r"(?:void )?([A-Za-z_0-9]+)(<.*?>)?\(", name + "(")
    if not m:
        return name
    base, targs = m.group(1), m.group(2) or ""
    if base.startswith("rs_") and "unsigned long" in targs:
        return base + "<u64>"
    if base in ("cmp_count", "cmp_write", "vscan_partial", "vscan_write"):
        f = re.search(r"([A-Za-z]+)>$", targs)
        return base + ("<%s>" % f.group(1) if f else "")
    return base


def latest(src, sub, pattern):
    files = sorted(glob.glob(os.path.join(src, sub, "*", pattern)), key=os.path.getmtime)
    return files[-1] if files else None


def size_classes(rows, grid_key):
    """max grid per kernel name: launches below half of it are the [small] class (only for the rs_* kernels)"""
    mx = {}
    for r in rows:
        k = short(r["Kernel_Name"])
        mx[k] = max(mx.get(k, 0), int(r[grid_key]))
    return mx


def klass(r, mx, grid_key):
    k = short(r["Kernel_Name"])
    if k.startswith("rs_") and int(r[grid_key]) * 2 < mx[k]:
        return k + "[small]"
    return k


def avg(v):
    return sum(v) / len(v) if v else 0.0


def read_pmc(src, sub, counters):
    f = latest(src, sub, "*counter_collection.csv")
    out = {}
    if not f:
        return out
    rows = list(csv.DictReader(open(f)))
    mx = size_classes(rows, "Grid_Size")
    for r in rows:
        if r["Counter_Name"] in counters:
            out.setdefault(klass(r, mx, "Grid_Size"), {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    cfg = sys.argv[2] if len(sys.argv) > 2 else "C3"
    src = os.path.join(ROOT, "gpurun_out", "prof_%s" % tag if cfg == "C3" else "prof_%s_%s" % (tag, cfg))
    if cfg != "C3":
        tag = "%s_%s" % (tag, cfg.lower())
    os.makedirs(DST, exist_ok=True)
    stats = latest(src, "trace", "*kernel_stats.csv")
    trace = list(csv.DictReader(open(latest(src, "trace", "*kernel_trace.csv"))))
    with open(os.path.join(src, "trace_bench.json")) as f:
        bench = json.loads(f.read().strip().splitlines()[-1])
    P = bench["roofline"]["seed_pass"]["positions"] / max(1, bench["roofline"]["seed_pass"].get("seed_passes_per_step", 1)) if cfg == "C4" else bench["roofline"]["seed_pass"]["positions"]
    mx = size_classes(trace, "Grid_Size_X")
    dur = {}
    for r in trace:
        dur.setdefault(klass(r, mx, "Grid_Size_X"), []).append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    # passes of the hot path in the process: a kernel that runs once per pass (the guide tree's run list on the progressive path, the assembly's fill otherwise)
    passes = len(dur.get("run_summary", []) or dur.get("as_fill", []) or dur.get("dp_step2", []) or dur.get("dp_step", [])) or 1
    total = sum(sum(v) for v in dur.values())
    fetch = read_pmc(src, "pmc_fetch", ("FETCH_SIZE",))
    write = read_pmc(src, "pmc_write", ("WRITE_SIZE",))
    lds = read_pmc(src, "pmc_lds", ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"))
    cal = 1.0
    if cfg == "C3":
        if "rs_hist" in fetch and P:
            cal = (4.0 * P) / (avg(fetch["rs_hist"]["FETCH_SIZE"]) * 1024.0)
    else:
        # the other configs run seed passes of many sizes (and 64-bit keys): no single launch with a known byte count.  The factor is a property of
        # the access pattern, not of the workload: the one calibrated on the C3 command of the same round is applied
        try:
            t3 = json.load(open(os.path.join(DST, "roofline_traffic.json")))
            cal = next(v["fetch_calibration"] for k, v in t3.items() if isinstance(v, dict) and "fetch_calibration" in v)
        except Exception:
            cal = 2.0                                  # MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of a wide coalesced streaming read
    traffic = {}
    cmd_text = ("Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --config " + cfg +
                "` (" + str(passes) + " passes of the hot path in the process: warmup, timed host-to-host, device-resident and HIP-event legs, all over BASELINE config " + cfg + ": " +
                bench["config"]["workload"][:70] + " ..., P = " + str(int(P)) + " windows per pass).  PMC passes: `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE` and `--pmc SQ_LDS_BANK_CONFLICT "
                "SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY`, separate runs (tools/profile_gpu.sh).  The raw `--stats` table is " + tag + "_kernel_stats.csv; the table below is built "
                "from the kernel trace of the same run so that the main sort and the ~20 small sorts per pass (tagged [small], split by grid size) are not averaged together.")
    lines = ["# rocprofv3 summary, round %s" % tag, "", cmd_text, "",
             "Kernel time per pass of the hot path: %.3f ms (sum of all kernel durations / %d passes)." % (total / passes / 1e6, passes), "",
             "| kernel | launches/pass | avg us | us/pass | share % | FETCH_SIZE KiB/launch (raw) | WRITE_SIZE KiB/launch | HBM bytes/launch (calibrated) |",
             "|---|---|---|---|---|---|---|---|"]
    order = sorted(dur.items(), key=lambda kv: -sum(kv[1]))
    for k, v in order:
        fr = avg(fetch.get(k, {}).get("FETCH_SIZE", []))
        wr = avg(write.get(k, {}).get("WRITE_SIZE", []))
        hbm = fr * 1024.0 * cal + wr * 1024.0
        traffic[k] = {"fetch_kib_raw": round(fr, 1), "write_kib": round(wr, 1), "fetch_calibration": round(cal, 3),
                      "hbm_bytes_per_launch": round(hbm), "avg_ns": round(avg(v), 1), "calls": len(v)}
        if sum(v) / total < 0.0015:
            continue
        lines.append("| %s | %.1f | %.1f | %.1f | %.2f | %.0f | %.0f | %.3g |" % (
            k, len(v) / passes, avg(v) / 1e3, sum(v) / passes / 1e3, 100.0 * sum(v) / total, fr, wr, hbm))
    lines += ["", "LDS and wait counters (averages per launch; bank-conflict rate = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, "
              "wait share = SQ_WAIT_ANY / SQ_WAVE_CYCLES):", "",
              "| kernel | SQ_LDS_IDX_ACTIVE | SQ_LDS_BANK_CONFLICT | conflict rate | SQ_WAVE_CYCLES | wait share |", "|---|---|---|---|---|---|"]
    for k, v in order[:16]:
        c = lds.get(k)
        if not c:
            continue
        ia, bc, wc, wa = (avg(c.get(n, [])) for n in ("SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"))
        lines.append("| %s | %.3g | %.3g | %s | %.3g | %s |" % (k, ia, bc, ("%.2f" % (bc / ia)) if ia else "-", wc,
                                                             ("%.2f" % (wa / wc)) if wc else "-"))
        traffic.setdefault(k, {}).update({"lds_idx_active": round(ia), "lds_bank_conflict": round(bc)})
    lines += ["", "FETCH_SIZE calibration factor (C3: 4*P bytes / raw FETCH_SIZE of the main sort's rs_hist launch; other configs: the C3 factor of the same round): %.3f "
              "(MI355X_MICROARCH.md gives 2x for coalesced streaming reads on gfx950; the measured factor is applied to every kernel's FETCH_SIZE)." % cal, "",
              "Bench line of the profiled run:", "", "```json", json.dumps(bench), "```"]
    with open(os.path.join(DST, "%s_summary.md" % tag), "w") as f:
        f.write("\n".join(lines) + "\n")
    # bench.py quotes roofline.traffic from this file only while the kernel sources are the ones that were profiled
    sys.path.insert(0, ROOT)
    import subprocess
    from bench import kernel_source_digest
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        commit = "?"
    traffic["_meta"] = {"round": tag, "commit": commit, "kernel_source_digest": kernel_source_digest(),
                        "command": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE --kernel-trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --config " + cfg}
    tpath = os.path.join(DST, "roofline_traffic.json")
    if cfg == "C3":
        try:
            keep = json.load(open(tpath)).get("_configs", {})        # the other configs' sections stay (each carries its own digest)
        except Exception:
            keep = {}
        traffic["_configs"] = keep
        out = traffic
    else:
        try:
            out = json.load(open(tpath))
        except Exception:
            out = {}
        out.setdefault("_configs", {})[cfg] = traffic
    with open(tpath, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    shutil.copy(stats, os.path.join(DST, "%s_kernel_stats.csv" % tag))
    for c, d in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
        with open(os.path.join(DST, "%s_%s_per_kernel.csv" % (tag, c.lower())), "w") as f:
            f.write("kernel,launches,avg_kib_raw\n")
            for k, v in sorted(d.items()):
                f.write("%s,%d,%.1f\n" % (k, len(v[c]), avg(v[c])))
    with open(os.path.join(DST, "%s_lds_counters_per_kernel.csv" % tag), "w") as f:
        f.write("kernel,launches,SQ_LDS_IDX_ACTIVE,SQ_LDS_BANK_CONFLICT,SQ_WAVE_CYCLES,SQ_WAIT_ANY\n")
        for k, c in sorted(lds.items()):
            f.write("%s,%d,%.0f,%.0f,%.0f,%.0f\n" % (k, len(c.get("SQ_WAVE_CYCLES", [])), avg(c.get("SQ_LDS_IDX_ACTIVE", [])),
                                                 avg(c.get("SQ_LDS_BANK_CONFLICT", [])), avg(c.get("SQ_WAVE_CYCLES", [])),
                                                 avg(c.get("SQ_WAIT_ANY", []))))
    print("\n".join(lines[:60]))


if __name__ == "__main__":
    main()
