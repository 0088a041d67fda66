#!/bin/bash
# round 6, GPU call 9: when the edges of the 12 Mb call begin and end (HX_DEBUG=2), the host-staged --gpus 2 rehearsal, the group code at N = 1
set -u
O=gpurun_out/r06_9
mkdir -p $O
export HASLR_BENCH_DIR=/tmp/haslr_bench
AB_WORKLOAD=yeast AB_PASSES=3 HX_DEBUG=2 timeout 600 python tools/dev_r05_ab.py - 2> $O/edgedump.err | grep RESULT
python - <<PY > $O/top_edges.txt
import re
E=[]
for line in open("$O/edgedump.err"):
    if not line.startswith("[hx-edge]"): continue
    t=line.split(); d={t[i]:float(t[i+1]) for i in range(2,len(t)-1,2)}; d["e"]=int(t[1]); E.append(d)
n=len(E)//3
E=E[-n:]   # the last pass
end=max(d["end_us"] for d in E)
print("edges",n,"last end ms",end/1e3)
for d in sorted(E,key=lambda d:-d["end_us"])[:14]:
    tot=sum(d[k] for k in ("decode","dp","tb","graph","order","csr"))
    print("edge %d lmax %d nseq %d cls %d lanes %d members %d begin %.1f ms end %.1f ms chain %.1f ms (dp %.1f tb %.1f graph %.1f csr %.1f) rows %d" % (d["e"],d["lmax"],d["nseq"],d["cls"],d["lanes"],d["members"],d["begin_us"]/1e3,d["end_us"]/1e3,tot/2.4e6,d["dp"]/2.4e6,d["tb"]/2.4e6,d["graph"]/2.4e6,d["csr"]/2.4e6,d["rows"]))
PY
cat $O/top_edges.txt
rm -f $O/edgedump.err
HASLR_BENCH_FORCE_GROUP=1 HASLR_GROUP_TRANSPORT=host timeout 600 python bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_group1.json 2> $O/bench_group1.err
grep "step " $O/bench_group1.err | cut -c1-100
HASLR_GROUP_TRANSPORT=host timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench_group2_host.json 2> $O/bench_group2_host.err
tail -3 $O/bench_group2_host.err | cut -c1-300
