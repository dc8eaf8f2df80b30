"""Regenerate the inputs of one sweep case without a GPU: dry_case.py <fuzz script> <seed> <it>  ->  prints the options, saves the genomes
(the script's own except-branch does both when the stand-in context raises)."""
import sys, os, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mauvealigner_amd import _lib
class Dry:
    def __init__(self, *a): pass
    def set_genomes(self, gs): self.gs = gs
    def seed_mums(self, pat, mode=0, extend=True, mask=0):
        from oracle import pyoracle as O
        return O.find_matches(self.gs, pat, mode=mode, extend=extend, mask=mask)
    def __getattr__(self, k):
        def f(*a, **kw): raise RuntimeError('dry run: ' + k)
        return f
_lib.Context = Dry
script, seed, it = sys.argv[1], sys.argv[2], sys.argv[3]
os.environ['FUZZ_IT0'] = it; os.environ['FUZZ_IT'] = it; os.environ['FUZZ_N'] = '1'
sys.argv = [script, seed, '100'] + sys.argv[4:]
runpy.run_path(script, run_name='__main__')
