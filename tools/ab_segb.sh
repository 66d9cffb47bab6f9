#!/bin/bash
for sb in 1 2 4 8; do
  echo "== segb $sb"
  FVDB_SEGB=$sb python tools/quick_ivf_bench.py 700000 1024 32 1024 | grep -E "QPS|fine_scan"
done
