#!/bin/bash
# A/B of the whole bench with traversal kernel variants (dev aid).  usage: ab_bench_graph.sh "<extra bench flags>" name=ENV=VAL ...
mkdir -p gpurun_out/ab
EXTRA="$1"; shift
for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  env $envs python bench.py --steps 100 --warmup 3 --nprobe 32 --ef 50 --no-cpu-baseline --compare-host-walk 0 $EXTRA > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.log
  python - <<PY
import json
j=json.load(open("gpurun_out/ab/$name.json"))
print("$name", j["value"], "q/s", j["ms_per_step"], "ms/step recall", j["config"]["recall_at_10"], "graph kernel ms", j["roofline"]["graph_traversal_kernel"]["kernel_ms"])
PY
done
