import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mauvealigner_amd import _lib, synth
ctx=_lib.Context(0)
gs=synth.make_config('C5', 1.0)
ctx.set_genomes(gs)
p=_lib.default_params()
ctx.align(p, fetch=False)
ctx.profile(True); ctx.profile_reset(); ctx.align(p, fetch=False); ctx.profile(False)
print({k: (round(v['ms'],3), v['launches']) for k,v in ctx.profile_get().items()})
t=time.perf_counter(); ctx.align(p, fetch=False); print('C5 %.1f ms' % ((time.perf_counter()-t)*1e3), ctx.stage_times())
