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


def test_random_small_entries():
    """Randomised checks of the smaller C-ABI entries: multiplicity masks over arbitrary genome subsets, sorted mer lists
    (incl. 64-bit keys and genomes shorter than the seed), SeedMatchEnumerator on repetitive genomes, dp_batch with
    sizes around the kernel's class boundaries.  (scratch-scale run of the same generator: > 50 000 cases.)"""
    from mauvealigner_amd import _lib
    ctx = _lib.Context(0)
    try:
        for it in range(240):
            rng = np.random.default_rng(7000003 + it)
            kind = it % 4
            if kind == 0:
                N = int(rng.integers(2, 7)); L = int(rng.integers(100, 3000))
                anc = rng.integers(0, 4, L, dtype=np.uint8)
                gs = [synth.mutate(anc, float(rng.choice([0.0, 0.02, 0.1])), rng) if rng.random() < 0.8 else rng.integers(0, 4, L, dtype=np.uint8)
                      for _ in range(N)]
                ctx.set_genomes(gs)
                pat = O.get_seed(int(rng.choice([5, 7, 9, 11, 15, 21])), int(rng.integers(0, 3)))
                for _ in range(3):
                    mask = int(rng.integers(0, 1 << N)); mode = int(rng.integers(0, 2)); ext = bool(rng.integers(0, 2))
                    a = ctx.seed_mums(pat, mode=mode, mask=mask, extend=ext); b = O.find_matches(gs, pat, mode=mode, mask=mask, extend=ext)
                    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (it, mask, mode, ext)
            elif kind == 1:
                N = int(rng.integers(1, 4))
                gs = [rng.integers(0, 4, int(rng.integers(1, 5000)), dtype=np.uint8) if rng.random() < 0.8
                      else np.tile(rng.integers(0, 4, 7, dtype=np.uint8), 300) for _ in range(N)]
                ctx.set_genomes(gs)
                pat = O.get_seed(int(rng.choice([5, 9, 13, 16, 17, 23, 31])), int(rng.integers(0, 3)))
                s = int(rng.integers(0, N))
                a = ctx.sorted_mer_list(s, pat); b = O.sorted_mer_list(gs[s], pat)
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (it, s)
            elif kind == 2:
                unit = rng.integers(0, 4, int(rng.integers(20, 300)), dtype=np.uint8)
                parts = [rng.integers(0, 4, int(rng.integers(10, 500)), dtype=np.uint8)]
                for _ in range(int(rng.integers(1, 6))):
                    parts.append(unit if rng.random() < 0.6 else synth.revcomp(unit))
                    parts.append(rng.integers(0, 4, int(rng.integers(1, 300)), dtype=np.uint8))
                g = np.concatenate(parts); ctx.set_genomes([g])
                args = (int(rng.integers(2, 4)), int(rng.choice([3, 10, 1000])), bool(rng.integers(0, 2)))
                pat = O.get_seed(int(rng.choice([7, 9, 11])), 0)
                a = ctx.seed_match_enumerate(0, pat, *args); b = O.seed_match_enumerate(g, pat, *args)
                assert all(np.array_equal(x, y) for x, y in zip(a, b)), (it, args)
            else:
                nseq = int(rng.integers(2, 9)); ivs = []
                for _ in range(int(rng.integers(1, 40))):
                    base = rng.integers(0, 4, int(rng.choice([3, 15, 16, 17, 31, 33, 64, 65, 127, 129, 200, 700])), dtype=np.uint8)
                    iv = []
                    for _ in range(nseq):
                        r = rng.random()
                        if r < 0.15:
                            iv.append(np.zeros(0, np.uint8))
                        elif r < 0.25:
                            iv.append(rng.integers(0, 4, int(rng.integers(1, 300)), dtype=np.uint8))
                        else:
                            iv.append(synth.mutate(base, float(rng.choice([0.0, 0.05, 0.3])), rng, indel_frac=0.3)[: int(rng.integers(1, len(base) + 1))])
                    ivs.append(iv)
                cols, score = ctx.dp_batch(ivs)
                for iv, c, s in zip(ivs, cols, score):
                    ec, es = O.align_interval(iv)
                    assert np.array_equal(c, ec) and int(s) == es, it
    finally:
        ctx.close()
