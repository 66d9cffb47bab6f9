#!/bin/bash
for w in 2 3 4 5 8; do
  echo "== wgs/cu $w"
  FVDB_SCAN_WGS_PER_CU=$w python tools/quick_ivf_bench.py 700000 1024 32 1024 | grep -E "fine_scan"
  FVDB_SCAN_WGS_PER_CU=$w python tools/flat_bench.py 400000 4096
done
