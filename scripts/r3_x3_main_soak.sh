#!/bin/bash
# ON THE GPU BOX: MainNetwork under split precision through the actor (parity with the oracle search) and a 40-move run
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_selfplay.py -x -q -s -k "mainnetwork" > gpurun_out/r3_x3_actor_tests.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r3_x3_actor_tests.log; grep -a "MainNetwork through\|passed\|failed\|rc=" gpurun_out/r3_x3_actor_tests.log | tail -5
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python bench.py --network transgo --dtype f32x3 --steps 40 --warmup 2 --no-cpu-baseline > gpurun_out/r3_transgo_x3_soak_line.json 2> gpurun_out/r3_transgo_x3_soak.err || { tail -20 gpurun_out/r3_transgo_x3_soak.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/r3_transgo_x3_soak_line.json').read().strip().splitlines()[-1])
print(d['value'], d['games_per_hour'], d['extra']['tree_errors'], d['extra']['truncated_tree_blocks'], d['selfplay_games'] if 'selfplay_games' in d else '')"
