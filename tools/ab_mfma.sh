#!/bin/bash
# development aid: GPU tests, then the IVF chain timing for a few MFMA variants
cd "$(dirname "$0")/.."
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -n 30 gpurun_out/gpu_tests.log; exit 1; }
tail -n 2 gpurun_out/gpu_tests.log
for v in "FVDB_MFMA_M=2" "FVDB_MFMA_M=1" "FVDB_MFMA_M=4" "FVDB_MFMA_M=2 FVDB_MFMA_WGS_PER_CU=2" "FVDB_MFMA_M=2 FVDB_MFMA_WGS_PER_CU=3"; do
  echo "=== $v"
  env $v timeout -k 10 200 python tools/ivf_bench2.py "$@" 2>&1 | tail -n 3
done
