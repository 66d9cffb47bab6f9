#!/bin/bash
# A/B the scan-kernel variants in one GPU session (dev aid)
for v in "$@"; do
  export FVDB_HIP_LIB=$PWD/fabstir-vectordb_amd/lib_variants/libfvdb_hip_$v.so
  echo "== variant $v"
  python tools/flat_bench.py 400000 4096
  python tools/quick_ivf_bench.py 700000 1024 32 1024 | grep -E "QPS|fine_scan"
done
