# development aid: sharing knobs in the many-edge regime (140 Mb data set): costliest edges shared x members per edge
cd $GRAFT_REPO_ROOT
python tools/full_size_check.py fly --no-oracle --no-identity --passes 1 > /dev/null 2>&1
for cfg in "$@"; do
  set -- $(echo $cfg | tr ':' ' ')
  echo "== TOPK=$1 CLUSTER_MAX=$2"; HX_POA_CLUSTER_TOPK=$1 HX_POA_CLUSTER_MAX=$2 python tools/full_size_check.py fly --no-oracle --no-identity --reuse 2>&1 | grep -E "gpu pass 1"
done
