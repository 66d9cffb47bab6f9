#!/bin/bash
# two ranks of the multi-GPU path on ONE GPU (gloo collectives through host copies): checks the orchestration,
# not the speed.  Each rank builds a small index so both fit comfortably.
cd $GRAFT_REPO_ROOT
export FVDB_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 \
  bench.py --gpus 2 --steps 5 --warmup 1 --n-vectors 200000 --nlist 256 --train-sample 50000 --no-cpu-baseline --compare-host-walk 0
