import os, random, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
from haslr_amd import host, hip
import ctypes as _C
class _Dummy:
    argtypes = None; restype = None
    def __call__(self, *a): return -1
class _Shim(_C.CDLL):
    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            if name.startswith('hx'):
                d = _Dummy(); self.__dict__[name] = d; return d
            raise
if os.environ.get("HASLR_DEV_LIBDIR"):
    hip._LIBDIR = os.environ["HASLR_DEV_LIBDIR"]; hip.C.CDLL = _Shim
import orclib
rnd = random.Random(7)
def noisy(t, ins=0.06, dele=0.04, sub=0.03):
    out = []
    for c in t:
        r = rnd.random()
        if r < dele: continue
        out.append(rnd.choice("ACGT") if r < dele + sub else c)
        while rnd.random() < ins: out.append(rnd.choice("ACGT"))
    return "".join(out)
ctx = hip.HipContext(0)
for L in [int(x) for x in os.environ.get('CASE_LENS', '5000,9000,17000').split(',')]:
    for nseq in (3, 6):
        tmpl = "".join(rnd.choice("ACGT") for _ in range(L))
        seqs = [noisy(tmpl) for _ in range(nseq)]
        want = orclib.poa_consensus(seqs)
        ob = os.environ.get('CASE_ONLY_BLOCK')
        for blk, dirb in (((int(ob), 0),) if os.environ.get('CASE_DIR0') else ((int(ob), 1), (int(ob), 0)) if ob else ((0, 1), (1024, 1), (1024, 0), (512, 1))):
            ctx.set_poa_block(blk); hip.lib().hx_set_poa_traceback(ctx._h, dirb)
            try:
                got = ctx.poa_sequences([seqs])[0]
                print('L', L, 'nseq', nseq, 'block', blk, 'dir', dirb, 'OK' if got == want else 'DIFF (%d vs %d)' % (len(got), len(want)), flush=True)
            except Exception as e:
                print('L', L, 'nseq', nseq, 'block', blk, 'dir', dirb, 'ERROR', str(e)[:100], flush=True)
