"""Randomised parity sweep on the GPU: random genome sets (2-6 genomes, substitutions, indels, inversions,
duplications, insertions, a reverse-complemented or near-empty member), random seed weight and rank, every finder
mode with and without extension, mauve_align under random options and the progressive aligner -- all bit-exact
against the CPU oracle.  (scratch-scale runs of the same generator: > 9000 small and > 500 large cases.)"""
import numpy as np
import pytest

from mauvealigner_amd import synth
from oracle import pyoracle as O

pytestmark = pytest.mark.gpu
KEYS = ("anchor_start", "anchor_length", "left", "right", "reverse", "col_off", "cols", "dp_score")


def _genomes(rng, lmax):
    N = int(rng.integers(2, 7))
    L = int(rng.integers(150, lmax))
    div = float(rng.choice([0.0, 0.01, 0.03, 0.08, 0.2]))
    anc = rng.integers(0, 4, L, dtype=np.uint8)
    gs = []
    for _ in range(N):
        x = synth.mutate(anc, div, rng, indel_frac=float(rng.choice([0.0, 0.1, 0.4])))
        r = rng.random()
        if r < 0.25 and len(x) > 100:
            a = int(rng.integers(0, len(x) - 50)); b = a + int(rng.integers(20, min(max(2000, lmax // 3), len(x) - a)))
            x = x.copy(); x[a:b] = synth.revcomp(x[a:b])
        elif r < 0.35 and len(x) > 100:
            a = int(rng.integers(0, len(x) - 50)); b = a + int(rng.integers(20, min(1500, len(x) - a)))
            x = np.concatenate([x, x[a:b]])
        elif r < 0.45:
            x = np.concatenate([x[: len(x) // 2], rng.integers(0, 4, int(rng.integers(1, max(800, lmax // 4))), dtype=np.uint8), x[len(x) // 2:]])
        elif r < 0.5:
            x = synth.revcomp(x)
        elif r < 0.53:
            x = x[: int(rng.integers(0, 30))]
        gs.append(np.ascontiguousarray(x))
    return gs


@pytest.mark.parametrize("seed,count,lmax", [(1, 120, 6000), (2, 8, 60000)])
def test_random_parity(seed, count, lmax):
    from mauvealigner_amd import _lib
    ctx = _lib.Context(0)
    try:
        for it in range(count):
            rng = np.random.default_rng(seed * 100003 + it)
            gs = _genomes(rng, lmax)
            N = len(gs)
            pat = O.get_seed(int(rng.choice([5, 7, 9, 11, 13, 15, 17, 21])), int(rng.integers(0, 3)))
            ctx.set_genomes(gs)
            for mode in (0, 1, 2):
                for ext in (True, False):
                    ln, st = ctx.seed_mums(pat, mode=mode, extend=ext)
                    eln, est = O.find_matches(gs, pat, mode=mode, extend=ext)
                    assert np.array_equal(ln, eln) and np.array_equal(st, est), (seed, it, mode, ext)
            ln, st = ctx.seed_mums(pat, mode=0, mask=(1 << N) - 1)
            eln, est = O.find_matches(gs, pat, mode=0, mask=(1 << N) - 1)
            assert np.array_equal(ln, eln) and np.array_equal(st, est), (seed, it, "masked")
            if min(len(g) for g in gs) < 40:
                continue
            kw = dict(seed_weight=int(rng.choice([0, 7, 9, 11])), mode=int(rng.integers(0, 2)), recursive=int(rng.integers(0, 2)),
                      collinear=int(rng.random() < 0.2), add_unaligned=int(rng.integers(0, 2)), extend_lcbs=int(rng.random() < 0.4))
            r = ctx.align(_lib.default_params(**kw))
            e = O.align(gs, O.default_params(**kw))["aln"]
            assert all(np.array_equal(r[k], e[k]) for k in KEYS), (seed, it, kw)
            if N >= 3 and it % 3 == 0:
                r = ctx.progressive_align(_lib.default_params())
                e = O.progressive_align(gs)["aln"]
                assert all(np.array_equal(r[k], e[k]) for k in ("left", "right", "reverse", "col_off", "cols", "dp_score")), (seed, it, "progressive")
    finally:
        ctx.close()


def test_device_canonical_sort_path():
    """Large candidate sets are put in canonical order on the device (canon_keys -> radix sort -> canon_gather).  The
    switch is a size threshold read once per process, so a child process with the threshold forced to 1 runs part of
    the sweep above through that path."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_gpu_fuzz as t\n"
            "t.test_random_parity(3, 40, 6000)\n"
            "print('CHILD OK')\n") % (root, os.path.join(root, "tests"))
    env = dict(os.environ, MAUVE_CANON_DEVICE_MIN="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "CHILD OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
