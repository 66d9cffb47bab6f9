#!/bin/bash
# Collects the evidence committed under profiles/ (run on the GPU box through gpurun; outputs in gpurun_out/final/).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final
rm -rf $O && mkdir -p $O
echo "== default bench"; python3 bench.py > $O/bench_c3_default.json 2> $O/bench_c3_default.log; tail -n 3 $O/bench_c3_default.log
echo "== one batch at a time"; python3 bench.py --no-cpu-baseline --compare-host-walk 0 --in-flight 1 > $O/bench_c3_in_flight_1.json 2> $O/bench_c3_in_flight_1.log; tail -n 1 $O/bench_c3_in_flight_1.log
echo "== tiny run (2 steps, no warmup)"; python3 bench.py --no-cpu-baseline --compare-host-walk 0 --nprobe 32 --ef 50 --steps 2 --warmup 0 2>&1 | tail -n 2 | cut -c1-120
echo "== sharded path forced on one rank"; FVDB_FORCE_SHARDED=1 python3 bench.py --steps 10 --compare-host-walk 0 --cpu-sample 256 > $O/bench_c3_forced_sharded.json 2> $O/bench_c3_forced_sharded.log; tail -n 2 $O/bench_c3_forced_sharded.log
echo "== rocprofv3 kernel trace of the bench command"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 bench.py --no-cpu-baseline --compare-host-walk 0 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.log
tail -n 1 $O/bench_under_rocprof.log
echo "== IVF chain alone: matrix-core path vs exact path"
python3 tools/ivf_bench2.py 700000 1024 32 1024 > $O/ivf_chain_mfma.log 2>&1; tail -n 3 $O/ivf_chain_mfma.log
FVDB_SCAN_EXACT=1 FVDB_COARSE_EXACT=1 python3 tools/ivf_bench2.py 700000 1024 32 1024 > $O/ivf_chain_exact.log 2>&1; tail -n 3 $O/ivf_chain_exact.log
echo "== traversal kernel alone"
for b in 1024 2048; do python3 tools/hnsw_dev_bench.py 300000 50 $b 2>&1 | tail -n 1; done | tee $O/graph_kernel.log
