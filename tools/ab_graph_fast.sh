#!/bin/bash
# A/B of the traversal kernels (dev aid): exact-heap kernel vs the fast kernel at R = 8 / 12 / 16 rows per round
mkdir -p gpurun_out/graph
FVDB_GRAPH_NO_FAST=1 python tools/graph_bench.py > gpurun_out/graph/old.log 2>&1 && tail -4 gpurun_out/graph/old.log
for r in 16 12 8; do
  FVDB_GRAPH_CHECK=1 FVDB_GRAPH_FAST_R=$r python tools/graph_bench.py > gpurun_out/graph/fast_r$r.log 2>&1 && tail -5 gpurun_out/graph/fast_r$r.log
done
