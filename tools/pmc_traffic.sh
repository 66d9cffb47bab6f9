#!/bin/bash
# HBM traffic of the list-scan kernel from PMC counters (separate passes, kernel-trace only), per launch.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --nprobe 48 --ef 50 --compare-host-walk 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py $ARGS > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections, json
out={}
for tag,ctr in (("fetch","FETCH_SIZE"),("write","WRITE_SIZE")):
    for f in glob.glob("gpurun_out/pmc_%s/*/*counter_collection.csv" % tag):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"]==ctr: agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k,v in agg.items():
            if "scan_topk_kernel<16, 1, 1>" in k or "hnsw_search_kernel" in k or "scan_topk_kernel<16, 1, 0>" in k:
                name = "list_scan" if "1, 1>" in k else ("coarse_scan" if "1, 0>" in k else "hnsw_search")
                out.setdefault(name,{})[ctr+"_KB_avg"]=sum(v)/len(v); out[name]["launches"]=len(v)
print(json.dumps(out, indent=1))
open("gpurun_out/pmc_traffic.json","w").write(json.dumps(out, indent=1))
PY
