import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import ctypes as C
from mauvealigner_amd import _lib, synth
import bench
gs, _, L = bench.make_workload("C3", 1.0, 0)
ctx = _lib.Context(0)
packed, lens = bench.pack_pinned(gs)
bufs = _lib.ResultBuffers()
p = _lib.default_params(seed_weight=15)
def step(tm):
    t0 = time.perf_counter(); ctx.set_genomes_packed(packed, lens); t1 = time.perf_counter()
    sz = _lib.AlignSizes(); ctx._chk(ctx.L.mauve_align(ctx.h, C.byref(p), C.byref(sz)), "align"); t2 = time.perf_counter()
    r = ctx._fetch(sz, None, False, bufs); t3 = time.perf_counter()
    tm[0] += t1 - t0; tm[1] += t2 - t1; tm[2] += t3 - t2
for _ in range(5): step([0, 0, 0])
tm = [0, 0, 0]; n = 20
for _ in range(n): step(tm)
print("upload %.3f ms, align %.3f ms, fetch %.3f ms" % tuple(x / n * 1e3 for x in tm))
# fetch pieces: python-side buffer lookup vs the C call
sz = _lib.AlignSizes(); ctx._chk(ctx.L.mauve_align(ctx.h, C.byref(p), C.byref(sz)), "align")
t = time.perf_counter()
for _ in range(20): ctx._fetch(sz, None, False, bufs)
print("re-fetch (tables on host already) %.3f ms" % ((time.perf_counter() - t) / 20 * 1e3))
