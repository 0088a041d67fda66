"""development aid: load a data set from text and from the index caches a CLI run left, time load / upload / chain for both"""
import os, sys, time, subprocess
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, ROOT)
from haslr_amd import hip, host
D = '/tmp/idxup'; os.makedirs(D, exist_ok=True)
glen = sys.argv[1] if len(sys.argv) > 1 else '60000000'
subprocess.check_call([ROOT + '/tools/hxsim', '--genome-len', glen, '--seed', '11', '--out-prefix', D + '/s'], stderr=subprocess.DEVNULL)
subprocess.check_call([ROOT + '/haslr_amd/bin/haslr_assemble', '-t', '32', '-c', D + '/s.contigs.fa', '-l', D + '/s.reads.fa', '-m', D + '/s.paf', '-d', D + '/out'], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL)
ctx = hip.HipContext(0)
for idx in (None, D + '/out', None, D + '/out'):
    t0 = time.perf_counter(); ds = host.Dataset(D + '/s.contigs.fa', D + '/s.reads.fa', D + '/s.paf', threads=32, index_dir=idx); t1 = time.perf_counter()
    ctx.upload(ds); t2 = time.perf_counter()
    r = host.Run(ds, ds.params(), ctx.backend(), None); r.chain(); t3 = time.perf_counter()
    print('index' if idx else 'text ', 'load %.2f upload %.2f chain %.2f' % (t1 - t0, t2 - t1, t3 - t2), 'hits', ds.hits.n, flush=True)
    r.close(); ds.close()
