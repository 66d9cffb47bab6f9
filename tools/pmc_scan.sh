#!/bin/bash
# PMC counters for the scan kernel (separate passes; kernel-trace only) — dev aid
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_$1
shift
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d ${OUT}_a -- "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d ${OUT}_b -- "$@" > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for tag in ("a","b"):
    for f in glob.glob("${OUT}_%s/*/*counter_collection.csv" % tag):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
        for k,v in agg.items():
            if "scan_topk" in k:
                print(tag, k, {c: round(x/cnt[(k,c)]) for c,x in v.items()})
PY
