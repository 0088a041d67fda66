import sys, time, os, subprocess, filecmp
import numpy as np
ROOT=os.environ.get('GRAFT_REPO_ROOT','/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests')
from haslr_amd import host, hip
import orclib
glen = sys.argv[1] if len(sys.argv)>1 else '300000'
seed = sys.argv[2] if len(sys.argv)>2 else '7'
blk = int(sys.argv[3]) if len(sys.argv)>3 else 256
os.makedirs('/tmp/gt', exist_ok=True)
subprocess.check_call([ROOT+'/tools/hxsim','--genome-len',glen,'--seed',seed,'--out-prefix','/tmp/gt/s'] + os.environ.get('HXSIM_EXTRA','').split())
ds = host.Dataset('/tmp/gt/s.contigs.fa','/tmp/gt/s.reads.fa','/tmp/gt/s.paf')
prm = ds.params()
be = orclib.OracleBackend(ds, 8)
t=time.time(); ro = host.Run(ds, prm, be.table, '/tmp/gt/orc'); ro.all(); print('oracle run', time.time()-t, ro.timings())
ctx = hip.HipContext(0); ctx.upload(ds); ctx.set_poa_block(blk); ctx.set_poa_traceback(int(os.environ.get('HX_DIR','1')))
t=time.time(); rg = host.Run(ds, prm, ctx.backend(), '/tmp/gt/hip'); rg.all(); print('hip run', time.time()-t, rg.timings(), ctx.timing())
def cmpd(a,b,name):
    bad=[k for k in a if not np.array_equal(a[k],b[k])]
    print(name, 'OK' if not bad else 'DIFF '+str(bad))
    return not bad
cmpd(ro.chain_out(), rg.chain_out(), 'chain')
cmpd(ro.edges_out(), rg.edges_out(), 'edges')
cmpd(ro.coords_out(), rg.coords_out(), 'coords')
co, cg = ro.cns_out(), rg.cns_out()
nd = sum(1 for x,y in zip(co,cg) if x!=y)
print('cns edges', len(co), 'differ', nd, ro.cns_stats(), rg.cns_stats())
for f in sorted(os.listdir('/tmp/gt/orc')):
    same = filecmp.cmp('/tmp/gt/orc/'+f, '/tmp/gt/hip/'+f, shallow=False)
    if not same: print(f, 'DIFF')
print('files compared')
