#!/bin/bash
# ON A ONE-GPU BOX: the N>1 path of bench.py with two ranks sharing the card ("gloo" carries the collectives; RCCL needs one GPU
# per rank).  bench.py launches its own ranks; this checks the sharding, the gather to rank 0, the max-over-ranks timing and the
# JSON line (n_gpus 2, ranks, roofline, cpu_baseline) -- not a scaling number.  tests/test_gpu_bench.py runs the same command.
set -e
export TRANSGO_DIST_BACKEND=gloo
python3 bench.py --gpus 2 --games 1024 --steps 3 --warmup 1 "$@"
