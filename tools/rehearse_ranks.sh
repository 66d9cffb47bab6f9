#!/bin/bash
# Several ranks of the contract bench on ONE GPU (hosted transport: the C code path of the sharded search with the two
# exchanges carried over gloo).  Orchestration rehearsal only — the ranks share the GPU, so the numbers are not scaling
# numbers.  $1 = ranks, $2 = weak|strong, $3 = hosted|rccl (rccl: both ranks land on the one device, RCCL refuses, and the
# bring-up's fallback to the hosted transport is what gets rehearsed)
N=${1:-2}; MODE=${2:-weak}; TR=${3:-hosted}
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus $N --steps 20 --warmup 2 --transport $TR --scaling $MODE --n-vectors 200000 --nlist 256 --query-batches 4 \
  --nprobe 16 --ef 50 --no-cpu-baseline --allow-hosted
