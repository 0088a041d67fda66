cd $GRAFT_REPO_ROOT
# development aid: few-edge vs many-edge launch shapes at sizes around the threshold (bench.py --genome-len, single GPU)
run() { python bench.py --genome-len $1 --steps 2 --warmup 1 --no-cpu-baseline --no-configs1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['config']['edges'], round(d['ms_per_step'],1), round(d['roofline']['gcups'],1))"; }
for g in ${@:-36000000 60000000}; do
  echo "== genome $g: defaults"; run $g
  echo "== genome $g: few-edge shape (4 columns, ring uncut, 16 x 192 shared)"; HX_POA_COLS=4 HX_POA_RING_KB=40 HX_POA_CLUSTER_TOPK=192 HX_POA_CLUSTER_MAX=16 run $g
  echo "== genome $g: many-edge shape (8 columns, ring 11, 8 x 32 shared)"; HX_POA_COLS=8 HX_POA_RING_KB=11 HX_POA_CLUSTER_TOPK=32 HX_POA_CLUSTER_MAX=8 run $g
done
