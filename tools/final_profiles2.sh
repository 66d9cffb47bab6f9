#!/bin/bash
# rocprofv3 passes at the fixed operating point, one batch at a time (no sweep, no overlapping launches: every launch of a
# kernel is the same work and its duration is its own)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final
mkdir -p $O
rm -rf $O/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 bench.py --no-cpu-baseline --compare-host-walk 0 --nprobe 32 --ef 50 --in-flight 1 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.log
tail -n 1 $O/bench_under_rocprof.log
tools/pmc_traffic.sh > $O/pmc_run.log 2>&1
tail -n 5 $O/pmc_run.log
