#!/bin/bash
# Round-3 evidence committed under profiles/ (run on the GPU box through gpurun; outputs in gpurun_out/r03/).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
echo "== default bench (the driver's command)"; python3 bench.py > $O/bench_c3_default.json 2> $O/bench_c3_default.log; tail -n 4 $O/bench_c3_default.log | cut -c1-300
echo "== c2 / c5"; python3 bench.py --config c2 > $O/bench_c2.json 2> $O/bench_c2.log; tail -n 2 $O/bench_c2.log | cut -c1-200
python3 bench.py --config c5 > $O/bench_c5.json 2> $O/bench_c5.log; tail -n 2 $O/bench_c5.log | cut -c1-200
echo "== rocprofv3 kernel trace of the bench command, 8 batches in flight"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace8 -o bench -- python3 bench.py --no-cpu-baseline --compare-host-walk 0 --insert-sample 0 > $O/bench_in_flight_8_under_rocprof.json 2> $O/bench_in_flight_8_under_rocprof.log
tail -n 1 $O/bench_in_flight_8_under_rocprof.log | cut -c1-200
echo "== the same, one batch at a time"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace1 -o bench -- python3 bench.py --no-cpu-baseline --compare-host-walk 0 --insert-sample 0 --in-flight 1 > $O/bench_in_flight_1_under_rocprof.json 2> $O/bench_in_flight_1_under_rocprof.log
tail -n 1 $O/bench_in_flight_1_under_rocprof.log | cut -c1-200
python3 - <<PY
# overlap summary of the in-flight-8 trace: launches, wall span, summed duration, queues, per kernel of the timed region
import csv, glob, collections, json
f = glob.glob("$O/trace8/*/*kernel_trace.csv") + glob.glob("$O/trace8/*kernel_trace.csv")
rows = list(csv.DictReader(open(f[0])))
def pick(name):
    r = [x for x in rows if name in x["Kernel_Name"]]
    r.sort(key=lambda x: int(x["Start_Timestamp"]))
    return r[-200:]  # the timed region's launches are the last ones of the run (200 steps)
out = {}
for name in ("hnsw_search_fast_kernel", "scan_mfma_wg_kernel", "coarse_gemm_kernel", "coarse_select_kernel", "select_kernel", "threshold_direct_kernel"):
    r = pick(name)
    if not r: continue
    t0 = min(int(x["Start_Timestamp"]) for x in r); t1 = max(int(x["End_Timestamp"]) for x in r)
    dur = sum(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in r)
    out[name] = {"launches": len(r), "wall_span_ms": round((t1 - t0) / 1e6, 3), "summed_duration_ms": round(dur / 1e6, 3),
                 "mean_resident": round(dur / max(t1 - t0, 1), 2), "mean_duration_us": round(dur / len(r) / 1e3, 1),
                 "queues": len(set(x.get("Queue_Id", "") for x in r))}
open("$O/bench_in_flight_8_overlap_summary.json", "w").write(json.dumps(out, indent=1))
print(json.dumps(out, indent=1))
PY
for t in trace8 trace1; do f=$(ls $O/$t/*/*kernel_stats.csv $O/$t/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/bench_${t}_kernel_stats.csv; rm -rf $O/$t; done
echo "== PMC traffic"
bash tools/pmc_traffic.sh; cp gpurun_out/pmc_traffic.json $O/pmc_traffic.json
for t in fetch write mfma busy; do f=$(ls gpurun_out/pmc_$t/*/*counter_collection.csv gpurun_out/pmc_$t/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python3 - <<PY
import csv, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open("$f")):
    agg[(r["Kernel_Name"][:110], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open("$O/pmc_${t}_per_kernel.csv", "w") as o:
    o.write("kernel,counter,launches,median,mean\n")
    for (k, c), v in sorted(agg.items()):
        v = sorted(v)
        o.write('"%s",%s,%d,%.1f,%.1f\n' % (k, c, len(v), v[len(v) // 2], sum(v) / len(v)))
PY
done
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma gpurun_out/pmc_busy
if [ "$1" != "all" ]; then du -sh gpurun_out; exit 0; fi
echo "== supplementary workloads"; python3 bench.py --supplementary --no-cpu-baseline --compare-host-walk 0 --insert-sample 0 --steps 50 > $O/bench_c3_supplementary.json 2> $O/bench_c3_supplementary.log; grep supplementary $O/bench_c3_supplementary.log | cut -c1-200
echo "== isotropic cliff"; python3 tools/iso_cliff.py > $O/isotropic_filter.log 2>&1; cat $O/isotropic_filter.log | cut -c1-250
echo "== sequential build: C1 shapes and 300K"
python3 tools/build_bench.py --n 10000 --d 384 --mode 0 --check > $O/build_c1_mixture.log 2>&1; tail -n 3 $O/build_c1_mixture.log | cut -c1-300
python3 tools/build_bench.py --n 10000 --d 384 --mode 0 --gen refbench --check > $O/build_c1_refbench.log 2>&1; tail -n 3 $O/build_c1_refbench.log | cut -c1-300
FVDB_BUILD_DEBUG=1 python3 tools/build_bench.py --n 300000 --d 384 --mode 0 --tail 2048 --tail-mode 0 2>&1 | grep -v "1 linked" > $O/build_300k_mixture.log; tail -n 3 $O/build_300k_mixture.log | cut -c1-300
echo "== insert kernels, traced"; bash tools/insert_trace.sh; cp gpurun_out/ins/insert_kernels_summary.json $O/insert_kernels_summary.json
cat $O/build_c1_mixture.log $O/build_c1_refbench.log $O/build_300k_mixture.log > $O/side_workloads_builds.log
du -sh gpurun_out
