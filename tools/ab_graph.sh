#!/bin/bash
# A/B graph-kernel variants: the host mirror links lib/libfvdb_hip.so by rpath, so swap the file itself
L=fabstir-vectordb_amd/lib/libfvdb_hip.so
cp $L /tmp/libfvdb_hip_orig.so
for v in "$@"; do
  cp fabstir-vectordb_amd/lib_variants/libfvdb_hip_$v.so $L
  echo "== $v"; python tools/hnsw_dev_bench.py 2>&1 | grep -E "stamps|device_traversal=True" | tail -3
done
cp /tmp/libfvdb_hip_orig.so $L
