#!/usr/bin/env python3
"""Digest gpurun_out/prof (tools/profile_gpu.sh) into profiles/: the rocprofv3 kernel-stats table, per-kernel
HBM traffic from the PMC passes, and profiles/roofline_traffic.json (read by bench.py for roofline.traffic).

PMC handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE come from
separate passes, are in KiB, and on gfx950 FETCH_SIZE under-reports wide coalesced streaming reads by 2x.  Our
loads are 4-16 B per lane, so the factor is calibrated on a kernel with a known byte count in the same access
pattern: rs_hist reads exactly 4*P key bytes and writes a negligible histogram.
Usage: python tools/summarize_profile.py <round-tag>   (e.g. r01)
"""
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")


def short(name):
    """Kernel name without arguments.  The radix-sort kernels keep their key type: the main sort of C2 runs on 32-bit
    keys (rs_scatter), the small canonical-order sort on 64-bit keys (rs_scatter<u64>) -- one figure for both would
    describe neither."""
    m = re.match(r"(?:void )?([A-Za-z_0-9]+)(<[^(]*>)?", name)
    if not m:
        return name
    base, targs = m.group(1), m.group(2) or ""
    if base.startswith("rs_") and "unsigned long" in targs:
        return base + "<u64>"
    return base


def read_pmc(dirname, counter):
    out = {}
    files = sorted(glob.glob(os.path.join(SRC, dirname, "*", "*counter_collection.csv")), key=os.path.getmtime)
    for f in files[-1:]:            # newest run only
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                out.setdefault(short(row["Kernel_Name"]), []).append(float(row["Counter_Value"]))
    return out


def avg(v):
    return sum(v) / len(v) if v else 0.0


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    os.makedirs(DST, exist_ok=True)
    stats = sorted(glob.glob(os.path.join(SRC, "trace", "*", "*kernel_stats.csv")), key=os.path.getmtime)[-1]
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(SRC, "bench.json")) as f:
        bench = json.loads(f.read().strip().splitlines()[-1])
    P = bench["roofline"]["seed_pass"]["positions"]
    fetch, write = read_pmc("pmc_fetch", "FETCH_SIZE"), read_pmc("pmc_write", "WRITE_SIZE")
    cal = 1.0
    if "rs_hist" in fetch and P:          # the 32-bit-key instance: the main sort's histogram reads exactly 4 P bytes
        cal = (4.0 * P) / (avg(fetch["rs_hist"]) * 1024.0)
    traffic = {}
    lines = ["# rocprofv3 summary, round %s" % tag, "",
             "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 --warmup 1 "
             "--no-cpu-baseline` (1 warmup + 3 timed + 3 HIP-event passes = 7 passes of the hot path over BASELINE config "
             "C2, 3 x 5 Mbp, w = 15, P = %d windows).  PMC passes: `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, separate "
             "runs (tools/profile_gpu.sh)." % P, "",
             "| kernel | calls | avg us | total % | FETCH_SIZE KiB/launch (raw) | WRITE_SIZE KiB/launch | HBM bytes/launch (calibrated) |",
             "|---|---|---|---|---|---|---|"]
    for r in rows:
        k = short(r["Name"])
        fr, wr = avg(fetch.get(k, [])), avg(write.get(k, []))
        hbm = fr * 1024.0 * cal + wr * 1024.0
        traffic[k] = {"fetch_kib_raw": round(fr, 1), "write_kib": round(wr, 1), "fetch_calibration": round(cal, 3),
                      "hbm_bytes_per_launch": round(hbm), "avg_ns": float(r["AverageNs"]), "calls": int(r["Calls"])}
        lines.append("| %s | %s | %.1f | %s | %.0f | %.0f | %.3g |" % (k, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                  r["Percentage"], fr, wr, hbm))
    lines += ["", "FETCH_SIZE calibration factor (4*P bytes / rs_hist raw FETCH_SIZE): %.3f "
              "(the guide's 2x applies to 16-B-per-lane streams; ours are 4-B-per-lane)." % cal, "",
              "Bench line of the same build (un-profiled run):", "", "```json", json.dumps(bench), "```"]
    with open(os.path.join(DST, "%s_summary.md" % tag), "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(DST, "roofline_traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1, sort_keys=True)
    shutil.copy(stats, os.path.join(DST, "%s_kernel_stats.csv" % tag))
    for c, d in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
        with open(os.path.join(DST, "%s_%s_per_kernel.csv" % (tag, c.lower())), "w") as f:
            f.write("kernel,launches,avg_kib_raw\n")
            for k, v in sorted(d.items()):
                f.write("%s,%d,%.1f\n" % (k, len(v), avg(v)))
    print("\n".join(lines[:22]))


if __name__ == "__main__":
    main()
