#!/bin/bash
# A/B whole-bench runs of library variants (fabstir-vectordb_amd/lib_variants/libfvdb_hip_<name>.so): the host mirror links
# lib/libfvdb_hip.so by rpath, so the file itself is swapped and put back
cd $GRAFT_REPO_ROOT
L=fabstir-vectordb_amd/lib/libfvdb_hip.so
cp $L /tmp/libfvdb_hip_orig.so
mkdir -p gpurun_out
for v in orig "$@" orig "$@"; do
  if [ $v = orig ]; then cp /tmp/libfvdb_hip_orig.so $L; else cp fabstir-vectordb_amd/lib_variants/libfvdb_hip_$v.so $L; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --compare-host-walk 0 --nprobe 32 --ef 50 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.log || { cp /tmp/libfvdb_hip_orig.so $L; exit 1; }
  echo "== $v: $(python -c "import json,sys; d=json.load(open('gpurun_out/ab_$v.json')); print(d['value'], d['ms_per_step'], d.get('graph_traversal_kernel', d.get('roofline',{})).get('avg_ms', ''))")"
done
cp /tmp/libfvdb_hip_orig.so $L
