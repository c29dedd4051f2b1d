#!/bin/bash
# ON A ONE-GPU BOX: the N>1 path of bench.py with two ranks sharing the card ("gloo" carries the collectives; RCCL needs one GPU
# per rank).  Checks the sharding, the gather to rank 0, the max-over-ranks timing and the JSON line -- not a scaling number.
set -e
export TRANSGO_DIST_BACKEND=gloo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py \
    --gpus 2 --games 1024 --steps 3 --warmup 1 "$@"
