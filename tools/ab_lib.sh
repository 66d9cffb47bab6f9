#!/bin/bash
# A/B of the whole bench across builds of the engine (dev aid).
# usage: ab_lib.sh name=variant[:FLAGS][@ENV=VAL,...] ...   variants are fabstir-vectordb_amd/lib_variants/libfvdb_hip_<variant>.so;
# each is copied over lib/libfvdb_hip.so in the box's scratch copy, because the host mirror links the engine by that name
mkdir -p gpurun_out/ab
for spec in "$@"; do
  name=${spec%%=*}; rest=${spec#*=}; envs=""
  [[ "$rest" == *@* ]] && envs=$(echo "${rest#*@}" | tr ',' ' ') && rest=${rest%%@*}
  v=${rest%%:*}; flags=""
  [[ "$rest" == *:* ]] && flags=${rest#*:}
  cp fabstir-vectordb_amd/lib_variants/libfvdb_hip_$v.so fabstir-vectordb_amd/lib/libfvdb_hip.so
  env $envs python bench.py --steps 100 --warmup 3 --nprobe 32 --ef 50 --no-cpu-baseline --compare-host-walk 0 --query-batches 8 $flags > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.log
  python - <<PY
import json
j=json.load(open("gpurun_out/ab/$name.json"))
r=j["roofline"]
print("$name", j["value"], "q/s", j["ms_per_step"], "ms/step | filter", r["stage_ms"].get("mfma_filter_kernel"), "fine", r["stage_ms"].get("fine_scan"), "| fallbacks", j["config"]["ivf_scan_fallbacks"])
PY
done
cp fabstir-vectordb_amd/lib_variants/libfvdb_hip_base.so fabstir-vectordb_amd/lib/libfvdb_hip.so
