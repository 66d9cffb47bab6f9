#!/bin/bash
# development aid: which earlier test makes test_short_lists fail
cd "$(dirname "$0")/.."
for sel in "short_lists" "overflow or short_lists" "duplicates or short_lists" "deleted or short_lists" "fp16 or short_lists" "matches_oracle or short_lists"; do
  echo "=== -k '$sel'"
  timeout -k 10 120 python -m pytest tests/test_gpu_scan_mfma.py -x -q -m gpu -k "$sel" 2>&1 | tail -n 3
done
