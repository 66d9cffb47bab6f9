#!/bin/bash
# Kernel trace of the device-resident insert at the headline size: the 300K-node sequential build and the 2048-insert
# sample of bench.py's insert_path object.  Run on the GPU box through gpurun; summary -> gpurun_out/ins/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ins
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ins_trace -o ins -- python3 bench.py --no-cpu-baseline --compare-host-walk 0 --steps 20 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.log
tail -n 2 $O/bench_under_rocprof.log | cut -c1-300
python3 - <<PY
import csv, glob, json
f = glob.glob("/tmp/ins_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "hnsw_insert" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def summ(rs):
    out = {}
    for name in ("hnsw_insert_search_kernel", "hnsw_insert_commit_kernel"):
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs if name in r["Kernel_Name"]]
        if d:
            d.sort()
            out[name] = {"launches": len(d), "total_ms": round(sum(d) / 1e6, 2), "mean_us": round(sum(d) / len(d) / 1e3, 1),
                         "median_us": round(d[len(d) // 2] / 1e3, 1), "p90_us": round(d[int(len(d) * 0.9)] / 1e3, 1), "max_us": round(d[-1] / 1e3, 1)}
    if rs:
        out["wall_ms"] = round((int(rs[-1]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"])) / 1e6, 2)
    return out
res = {"what": "every launch of the device-resident insert in one bench.py run: the 300K-node sequential build of the headline's graph (no insert sample: --no-cpu-baseline)", "all": summ(rows)}
open("$O/insert_kernels_summary.json", "w").write(json.dumps(res, indent=1))
print(json.dumps(res, indent=1))
PY
f=$(find /tmp/ins_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/insert_bench_kernel_stats.csv
