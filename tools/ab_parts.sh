#!/bin/bash
# Where a step's time goes (dev aid): the bench with only one part of the hybrid search, and traversal tile variants.
# usage: ab_parts.sh name=FLAGS[@ENV=VAL,...] ...   e.g.  ivf="--parts historical"  r8="--parts recent@FVDB_GRAPH_FAST_R=8"
mkdir -p gpurun_out/ab
for spec in "$@"; do
  name=${spec%%=*}; rest=${spec#*=}; flags=${rest%%@*}; envs=""
  [[ "$rest" == *@* ]] && envs=$(echo "${rest#*@}" | tr ',' ' ')
  env $envs python bench.py --steps 100 --warmup 3 --nprobe 32 --ef 50 --no-cpu-baseline --compare-host-walk 0 --query-batches 8 $flags > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.log
  python - <<PY
import json
j=json.load(open("gpurun_out/ab/$name.json"))
r=j["roofline"]
print("$name", j["value"], "q/s", j["ms_per_step"], "ms/step | graph", r["kernel_ms"], "alone", r.get("kernel_ms_alone"), "| host", j["config"].get("host_collect_merge_ms_per_step"))
PY
done
