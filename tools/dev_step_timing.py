import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES","8")
ROOT=os.environ.get('GRAFT_REPO_ROOT','/root/repo'); sys.path.insert(0,ROOT)
import torch; torch.cuda.is_available()
import bench
from haslr_amd import hip, host
pre = bench.make_dataset(4600000, bench.SEED, "gpu")
ds = host.Dataset(pre+".contigs.fa", pre+".reads.fa", pre+".paf"); prm = ds.params()
ctx = hip.HipContext(0); ctx.upload(ds)
for it in range(3):
    t0=time.perf_counter(); run = host.Run(ds, prm, ctx.backend(), None); t1=time.perf_counter()
    run.chain(); t2=time.perf_counter(); run.graph(); t3=time.perf_counter(); run.coords(); t4=time.perf_counter(); run.consensus(); t5=time.perf_counter()
    run.close(); t6=time.perf_counter()
    print(f"create {t1-t0:.3f} chain {t2-t1:.3f} graph {t3-t2:.3f} coords {t4-t3:.3f} cons {t5-t4:.3f} close {t6-t5:.3f}", run and None, flush=True)
