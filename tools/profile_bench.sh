#!/bin/bash
# rocprofv3 --kernel-trace --stats of the contract bench (program directly after `--`).  $1 = tag, rest = bench flags
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --compare-host-walk 0 "$@" > $O/bench.json 2> $O/bench.log
ls $O | head; f=$(find $O -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
