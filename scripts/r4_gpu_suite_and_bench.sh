#!/bin/bash
# One gpurun call: the whole GPU suite, then the driver's bench command.  Steps are joined by && (a failed GPU step starts no other).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r4
mkdir -p $OUT
python -m pytest tests -m gpu -x -q -s --durations=15 2>&1 | grep -v "Invalid move" | tee $OUT/gpu_tests.log | tail -n 40 &&
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err &&
tail -c 3000 $OUT/bench_line.json
