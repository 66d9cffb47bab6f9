#!/bin/bash
# A/B of the whole bench under environment variants (dev aid).  usage: ab_bench.sh "<extra bench flags>" name=ENV=VAL[,ENV=VAL] ...
mkdir -p gpurun_out/ab
EXTRA="$1"; shift
for spec in "$@"; do
  name=${spec%%=*}; envs=$(echo "${spec#*=}" | tr ',' ' ')
  env $envs python bench.py --steps 100 --warmup 3 --nprobe 32 --ef 50 --no-cpu-baseline --compare-host-walk 0 --query-batches 8 $EXTRA > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.log
  python - <<PY
import json
j=json.load(open("gpurun_out/ab/$name.json"))
r=j["roofline"]
print("$name", j["value"], "q/s", j["ms_per_step"], "ms/step | graph", r["kernel_ms"], "ms | stages", r["stage_ms"])
PY
done
