import os, sys, subprocess
ROOT=os.environ.get('GRAFT_REPO_ROOT','/root/repo'); sys.path.insert(0,ROOT)
from haslr_amd import hip, host
os.makedirs('/tmp/fc',exist_ok=True)
subprocess.check_call([ROOT+'/tools/hxsim','--genome-len','150000','--seed','34','--cov','40','--gap-median','2500','--out-prefix','/tmp/fc/s'],stderr=subprocess.DEVNULL)
ds=host.Dataset('/tmp/fc/s.contigs.fa','/tmp/fc/s.reads.fa','/tmp/fc/s.paf')
ctx=hip.HipContext(0); ctx.upload(ds)
for fr in (None,'0','2'):
    if fr is None: os.environ.pop('HX_POA_FAR_ROWS',None)
    else: os.environ['HX_POA_FAR_ROWS']=fr
    print('FAR_ROWS',fr,file=sys.stderr,flush=True)
    r=host.Run(ds,ds.params(),ctx.backend(),None); r.all(); r.close()
