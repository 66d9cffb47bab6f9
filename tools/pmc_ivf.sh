#!/bin/bash
# SQ instruction / wait counters of the IVF chain's kernels under the bench (dev aid)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmci
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY \
   -d $R/gpurun_out/pmci/a -o pmc --output-format csv -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --compare-host-walk 0 --nprobe 32 --ef 50 --query-batches 4 --in-flight 1 --parts historical > $R/gpurun_out/pmci/a.json 2> $R/gpurun_out/pmci/a.log
python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/pmci/a/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in f:
    for row in csv.DictReader(open(fn)):
        k = row["Kernel_Name"][:48]
        if not any(t in k for t in ("scan_mfma", "coarse_", "select_kernel", "threshold_direct", "merge_topk")): continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k in acc:
    print(k, "dispatches", n[k], {c: round(v_ / max(n[k],1)) for c, v_ in acc[k].items()})
PY
