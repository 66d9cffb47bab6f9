#!/bin/bash
# SQ instruction / wait counters of the traversal kernels (dev aid): one --pmc pass per variant
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmcg
for v in old fast; do
  if [ $v = old ]; then export FVDB_GRAPH_NO_FAST=1; else unset FVDB_GRAPH_NO_FAST; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY \
     -d $R/gpurun_out/pmcg/$v -o pmc --output-format csv -- python3 $R/tools/graph_bench.py > $R/gpurun_out/pmcg/$v.log 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/pmcg/$v/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in f:
    for row in csv.DictReader(open(fn)):
        k = row["Kernel_Name"][:60]
        if "hnsw_search" not in k: continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); 
        if row["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k in acc:
    print("$v", k, "dispatches", n[k], {c: round(v_ / max(n[k],1)) for c, v_ in acc[k].items()})
PY
done
