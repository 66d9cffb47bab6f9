#!/bin/bash
# PMC counters of the hot kernels under the default bench command, per launch (separate passes, kernel-trace only):
#   FETCH_SIZE / WRITE_SIZE   bytes beyond L2 (gfx950: FETCH_SIZE counts wide streaming reads at half their bytes)
#   SQ_VALU_MFMA_BUSY_CYCLES  matrix-core busy cycles, against SQ_BUSY_CU_CYCLES / GRBM_GUI_ACTIVE
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--steps 20 --warmup 2 --no-cpu-baseline --compare-host-walk 0 --insert-sample 0 --nprobe 32 --ef 50 --query-batches 8"
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma gpurun_out/pmc_busy
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py $ARGS > gpurun_out/pmc_bench.json 2> gpurun_out/pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d gpurun_out/pmc_mfma -- python3 bench.py $ARGS > /dev/null 2> gpurun_out/pmc_mfma.log
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d gpurun_out/pmc_busy -- python3 bench.py $ARGS > /dev/null 2> gpurun_out/pmc_busy.log
python3 - <<PY
import csv, glob, collections, json
names = {"scan_mfma_wg_kernel": "list_scan", "scan_mfma_kernel<2, 1, 0>": "list_scan_wave_form", "scan_mfma_kernel<2, 1, 1>": "threshold_pass",
         "threshold_direct_kernel": "threshold", "hnsw_search_fast_kernel": "graph_traversal", "coarse_gemm_kernel": "coarse_gemm",
         "coarse_select_kernel": "coarse_select", "select_kernel<": "select", "merge_topk_kernel": "merge", "plan_": "plan",
         "prep_queries_kernel": "prep_queries", "hybrid_merge": "hybrid_merge"}
out = {}
for tag in ("fetch", "write", "mfma", "busy"):
    for f in glob.glob("gpurun_out/pmc_%s/*/*counter_collection.csv" % tag) + glob.glob("gpurun_out/pmc_%s/*counter_collection.csv" % tag):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, ctr), v in agg.items():
            for pat, name in names.items():
                if pat in k:
                    # bench-time launches only matter on average: the sweep and warmup launches are the same kernel
                    v = sorted(v)
                    out.setdefault(name, {})[ctr + ("_KB_avg" if ctr.endswith("_SIZE") else "_avg")] = v[len(v) // 2]  # median launch
                    out[name]["launches"] = len(v)
try:
    b = json.loads(open("gpurun_out/pmc_bench.json").read().strip().splitlines()[-1])
    out["nprobe"] = b["config"]["nprobe"]
    out["n_vectors"] = b["config"]["n_vectors"]
    out["command"] = "python3 bench.py $ARGS"
except Exception as e:
    out["nprobe"] = None
print(json.dumps(out, indent=1))
open("gpurun_out/pmc_traffic.json", "w").write(json.dumps(out, indent=1))
PY
