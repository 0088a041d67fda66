"""development aid: GPU path vs oracle over many random small data sets and launch shapes (consensus strings + assembly must be identical)"""
import os
import random
import subprocess
import sys

ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
sys.path.insert(0, ROOT + '/tests')
from haslr_amd import host, hip  # noqa: E402
import orclib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
if os.environ.get('FUZZ_LIBDIR'):   # a variant build of libhaslr_hip.so (tools/dev_variant.sh)
    hip._LIBDIR = os.path.join(ROOT, os.environ['FUZZ_LIBDIR'])
only = set(int(x) for x in os.environ['FUZZ_ONLY'].split(',')) if os.environ.get('FUZZ_ONLY') else None   # run these cases of the sequence only (the others still draw their random numbers)
ctx = hip.HipContext(0)
os.makedirs('/tmp/fz', exist_ok=True)
bad = 0
KNOBS = ('HX_POA_CLUSTER_MIN', 'HX_POA_MEMBER_LANES', 'HX_POA_CLUSTER_COLS', 'HX_POA_CLUSTER_MAX', 'HX_POA_MAX_INDEG', 'HX_POA_WAVE_MAX', 'HX_POA_FAR_ROWS', 'HX_POA_BATCHES', 'HX_POA_NODE_EST_PCT', 'HX_POA_RING_KB', 'HX_POA_COLS', 'HX_POA_CLUSTER_TOPK', 'HX_POA_SLOTS', 'HX_POA_WORKSPACE_GB', 'HX_POA_FORCE_CM', 'HX_POA_RING_ZERO', 'HX_POA_WIDE_MEMBERS', 'HX_POA_PRUNE', 'HX_POA_PRUNE_LANES', 'HX_POA_PASS_LANES', 'HX_POA_PRUNE_LAZY', 'HX_POA_CHAIN_MS', 'HX_POA_COLS2_TOP', 'HX_POA_PRUNE_SHARED', 'HX_POA_SLOTS_BY_WORK', 'HX_POA_BUCKET_HALF_OCTAVES', 'HX_POA_ORDER_BY_CELLS', 'HX_POA_FAR_SHIFT', 'HX_POA_OWN_BUCKET_FIRST', 'HX_POA_RESIDENT_FIRST')
for it in range(n):
    big = os.environ.get('FUZZ_BIG') is not None   # long gaps at real sizes: the default launch shapes (512-lane cluster members) get real work
    glen = rng.choice([400000, 700000, 1000000]) if big else rng.choice([40000, 60000, 90000, 150000, 250000])
    args = ['--genome-len', str(glen), '--seed', str(rng.randrange(1, 10**6)), '--model', rng.choice(['pacbio', 'nanopore', 'pacbio']), '--cov', str(rng.choice([8, 15, 25, 40, 70])),
            '--variant-per-mb', str(rng.choice([0, 5, 30])), '--gap-median', str(rng.choice([2500, 4000, 6000]) if big else rng.choice([300, 600, 1500, 3000])), '--out-prefix', '/tmp/fz/s']
    if rng.random() < 0.3:
        args[-2:-2] = ['--hairpin-frac', str(rng.choice([0.02, 0.1]))]
    skip = only is not None and it not in only
    if not skip:
        subprocess.check_call([ROOT + '/tools/hxsim'] + args, stderr=subprocess.DEVNULL)
    env = {}
    shape = rng.choice(['default', 'default', 'small-members', 'block']) if big else rng.choice(['default', 'small-members', 'one-wave', 'block'])
    for k in KNOBS:
        ctx.set_option(k, None)          # back to the default (the knobs are options of the context: hx_set_option)
    ctx.set_poa_block(0)
    if shape == 'small-members':
        env = {'HX_POA_CLUSTER_MIN': str(rng.choice([200, 400, 800])), 'HX_POA_MEMBER_LANES': str(rng.choice([64, 128, 256])),
               'HX_POA_CLUSTER_COLS': str(rng.choice([2, 4, 8])), 'HX_POA_CLUSTER_MAX': str(rng.choice([2, 3, 8, 16]))}
    elif shape == 'one-wave':
        env = {'HX_POA_WAVE_MAX': str(rng.choice([128, 256, 2048])), 'HX_POA_MAX_INDEG': str(rng.choice([2, 3, 16]))}
    elif shape == 'block':
        ctx.set_poa_block(rng.choice([64, 128, 256, 512, 1024]))
    if rng.random() < 0.3:
        env['HX_POA_FAR_ROWS'] = str(rng.choice([0, 1, 4, 16]))
    if rng.random() < 0.3:
        env['HX_POA_BATCHES'] = str(rng.choice([2, 3, 7]))
    if rng.random() < 0.3:
        env['HX_POA_RING_KB'] = str(rng.choice([1, 3, 7, 11]))      # small LDS rings: many far rows, rings without kept rows
    if rng.random() < 0.3:
        env['HX_POA_COLS'] = str(rng.choice([4, 8, 16]))
    if rng.random() < 0.2:
        env['HX_POA_CLUSTER_TOPK'] = str(rng.choice([0, 2, 1000]))
    if rng.random() < 0.4:
        env['HX_POA_SLOTS'] = str(rng.choice([1, 2, 3, 7]))          # persistent workgroups: a few workspace slots per launch class, many edges each (round 3)
    if rng.random() < 0.15:
        env['HX_POA_WORKSPACE_GB'] = str(rng.choice([1, 2, 4]))   # a small workspace cap: fewer slots, or several batches
    if rng.random() < 0.2:
        env['HX_POA_FORCE_CM'] = str(rng.choice([16, 32]))          # a wider kernel instance than the gap lengths ask for (spill-heavy row loops)
    if rng.random() < 0.1:
        env['HX_POA_RING_ZERO'] = '1'                               # no LDS ring: every kept row is read back from HBM
    if rng.random() < 0.5:
        env['HX_POA_WIDE_MEMBERS'] = str(rng.choice([0, 1, 100]))   # shared edges with 1024-lane members (default: the 4 costliest below 3 000 edges per call)
    if rng.random() < 0.6:
        env['HX_POA_PRUNE'] = str(rng.choice([80, 95, 95, 100, 104, 115]))   # exact score-bound pruning (round 5): thresholds below / at / above the previous alignment's score per base (above: repeats)
        env['HX_POA_PRUNE_LANES'] = str(rng.choice([64, 128, 128, 256]))
        env['HX_POA_PASS_LANES'] = str(rng.choice([0, 64, 64, 128, 256, -1, -1, -1]))     # column passes: the columns of a sequence window by window in a narrow workgroup (-1: width by estimated chain time)
        if env['HX_POA_PASS_LANES'] == '-1':
            env['HX_POA_CHAIN_MS'] = str(rng.choice([-1, 1, 3, 10, 40]))      # ... under a cap that small data sets reach: every width from 64 to 1024 lanes gets edges
        env['HX_POA_PRUNE_LAZY'] = str(rng.choice([0, 1, 1]))
        if 'HX_POA_WAVE_MAX' not in env and rng.random() < 0.7:
            env['HX_POA_WAVE_MAX'] = str(rng.choice([64, 128, 256]))          # several waves per workgroup on the short gaps of a small data set
    # round 6: 2 columns per lane for the longest chains, exact pruning inside shared edges (members run DP attempts), the slot policies of the need buckets, the launch order
    if rng.random() < 0.4:
        env['HX_POA_COLS2_TOP'] = str(rng.choice([0, 1, 4, 1000]))
    if rng.random() < 0.35:
        env['HX_POA_PRUNE_SHARED'] = str(rng.choice([90, 95, 104, 112]))
    if rng.random() < 0.3:
        env['HX_POA_SLOTS_BY_WORK'] = str(rng.choice([0, 1]))
        env['HX_POA_BUCKET_HALF_OCTAVES'] = str(rng.choice([0, 1]))
    if rng.random() < 0.2:
        env['HX_POA_ORDER_BY_CELLS'] = '1'
    if rng.random() < 0.25:
        env['HX_POA_OWN_BUCKET_FIRST'] = str(rng.choice([0, 1]))
    if rng.random() < 0.15:
        env['HX_POA_RESIDENT_FIRST'] = str(rng.choice([1, 2, 3]))
    if rng.random() < 0.15:
        env['HX_POA_FAR_SHIFT'] = str(rng.choice([2, 4, 6]))
    if rng.random() < 0.25:
        env['HX_POA_NODE_EST_PCT'] = str(rng.choice([2, 10, 30, 60]))
    pk = dict(min_aln_block=rng.choice([250, 500, 500, 1000]), min_aln_sim=rng.choice([0.8, 0.85, 0.85, 0.9]), min_edge_sup=rng.choice([2, 3, 3, 5]), max_uniq_dev=rng.choice([0.15, 0.15, 0.3]))
    if skip:
        continue
    if only is not None:
        print(it, 'RUN', ' '.join(args[:-2]), shape, env, pk, flush=True)
    ctx.set_options(**env)
    ds = host.Dataset('/tmp/fz/s.contigs.fa', '/tmp/fz/s.reads.fa', '/tmp/fz/s.paf')
    env = dict(env, **{k: str(v) for k, v in pk.items()})   # (printed with the knobs)
    be = orclib.OracleBackend(ds, 16)
    ro = host.Run(ds, ds.params(**pk), be.table, None)
    ro.all()
    ctx.upload(ds)
    rg = host.Run(ds, ds.params(**pk), ctx.backend(), None)
    try:
        rg.all()
    except host.HostError as e:
        if 'HX_POA_WORKSPACE_GB' in env and 'more POA workspace' in str(e):
            print(it, 'workspace cap too small for the largest edge (loud error, as it should be): again without the cap', flush=True)
            ctx.set_option('HX_POA_WORKSPACE_GB', None)
        elif shape == 'block' and 'needs the shared (cluster) mode' in str(e):
            print(it, 'a forced block size cannot hold a gap above 32 767 bases (loud error, as it should be): again with the automatic shape', flush=True)
            ctx.set_poa_block(0)
        else:
            raise
        rg.close()
        rg = host.Run(ds, ds.params(**pk), ctx.backend(), None)
        rg.all()
    ok = ro.cns_out() == rg.cns_out() and ro.assembly_fasta() == rg.assembly_fasta()
    print(it, 'OK' if ok else 'DIFF', ' '.join(args[:-2]), shape, env, 'edges', rg.n_edges, flush=True)
    bad += not ok
    rg.close(); ro.close(); be.close(); ds.close()
print('fuzz done, failures:', bad)
