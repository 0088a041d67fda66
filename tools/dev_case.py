"""development aid: ONE data set through the GPU path under several knob sets, each in a process of its own (a memory fault ends only that one);
usage: dev_case.py '<hxsim args>' block pk-json 'K=V,K=V' ['K=V,...' ...]   ("-" = no knobs)"""
import json
import os
import subprocess
import sys

ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
if len(sys.argv) > 1 and sys.argv[1] == '--child':
    sys.path.insert(0, ROOT)
    sys.path.insert(0, ROOT + '/tests')
    from haslr_amd import host, hip
    if os.environ.get('FUZZ_LIBDIR'):
        hip._LIBDIR = os.path.join(ROOT, os.environ['FUZZ_LIBDIR'])
    block, pk, spec = int(sys.argv[2]), json.loads(sys.argv[3]), sys.argv[4]
    env = dict(kv.split('=') for kv in spec.split(',')) if spec != '-' else {}
    ctx = hip.HipContext(0)
    ctx.set_poa_block(block)
    ctx.set_options(**env)
    ds = host.Dataset('/tmp/fz/c.contigs.fa', '/tmp/fz/c.reads.fa', '/tmp/fz/c.paf')
    ctx.upload(ds)
    rg = host.Run(ds, ds.params(**pk), ctx.backend(), None)
    try:
        rg.all()
        import hashlib
        h = hashlib.sha256()
        for c in rg.cns_out():
            h.update(c if isinstance(c, bytes) else str(c).encode()); h.update(b'\n')
        print('DONE edges', rg.n_edges, 'consensus', h.hexdigest()[:16], flush=True)
    except host.HostError as e:
        print('HOSTERROR', str(e)[:300], flush=True)
    sys.exit(0)
os.makedirs('/tmp/fz', exist_ok=True)
subprocess.check_call([ROOT + '/tools/hxsim'] + sys.argv[1].split() + ['--out-prefix', '/tmp/fz/c'], stderr=subprocess.DEVNULL)
for spec in sys.argv[4:]:
    r = subprocess.run(['timeout', '120', sys.executable, __file__, '--child', sys.argv[2], sys.argv[3], spec], capture_output=True, text=True)
    tail = [l for l in (r.stdout + r.stderr).splitlines() if l.startswith(('DONE', 'HOSTERROR', 'Memory access'))]
    print(f'[{spec}] block {sys.argv[2]} rc {r.returncode}:', ' | '.join(tail)[:400], flush=True)
    if r.returncode != 0 and os.environ.get('HX_DEBUG'):
        print('\n'.join(l[:300] for l in (r.stdout + r.stderr).splitlines() if l.startswith('[hx]'))[-6000:], flush=True)
