"""One CPU-baseline worker: aligns a synthetic config with the oracle and prints the seconds it took.
Test infrastructure (bench.py's cpu_baseline leg starts several of these to measure the all-cores figure);
never imported by the product.  usage: python -m oracle.cpu_worker <config> <scale> <seed_weight; 0 = the default weight>"""
import sys
import time


def main():
    cfg, scale, weight = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
    from mauvealigner_amd import synth
    from oracle import pyoracle as O
    gs = synth.make_config(cfg, scale=scale)
    print("ready", flush=True)
    sys.stdin.readline()                       # start line: all workers begin together
    t0 = time.perf_counter()
    kw = dict(seed_weight=weight) if weight > 0 else {}
    if cfg == "C4":                            # the progressiveMauve path at its call site's defaults (bench.py: params_for)
        O.progressive_align(gs, O.default_progressive_params(**kw))
    else:
        O.align(gs, O.default_params(**kw))
    print("%.6f %d" % (time.perf_counter() - t0, sum(len(g) for g in gs)), flush=True)


if __name__ == "__main__":
    main()
