import sys, os, subprocess
ROOT=os.environ.get('GRAFT_REPO_ROOT','/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests')
from haslr_amd import host, hip
import orclib
os.makedirs('/tmp/gt', exist_ok=True)
subprocess.check_call([ROOT+'/tools/hxsim','--genome-len','150000','--seed','31','--variant-per-mb','15','--out-prefix','/tmp/gt/c'], stderr=subprocess.DEVNULL)
ds = host.Dataset('/tmp/gt/c.contigs.fa','/tmp/gt/c.reads.fa','/tmp/gt/c.paf')
prm = ds.params()
be = orclib.OracleBackend(ds, 8)
ro = host.Run(ds, prm, be.table, None); ro.all()
ctx = hip.HipContext(0); ctx.upload(ds)
rg = host.Run(ds, prm, ctx.backend(), None); rg.all()
print('equal', ro.cns_out() == rg.cns_out(), rg.cns_stats())
