#!/bin/bash
# gpurun call A of the round's final evidence: the whole GPU suite, the driver's bench command, a 100-move run of C2 on the default
# pool (split-precision network: the tree behaves the same, the run is 2.7x shorter), C5's shape with 1024 boards for the memory figure.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r4
mkdir -p $OUT
B="python3 bench.py --no-launcher --no-cpu-baseline"
python -m pytest tests -m gpu -x -q -s --durations=12 2>&1 | grep -v "Invalid move" | tee $OUT/gpu_tests_final.log | tail -n 22 &&
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line_final.json 2> $OUT/bench_final.err &&
python3 -c "import json;l=json.load(open('$OUT/bench_line_final.json'));print('driver line', l['value'], l['roofline']['frac'], l['roofline_tree']['tree_ms_per_wave'], l['extra']['tree_pool'], {k:(v.get('value'),v.get('leg_wall_s')) for k,v in l['secondary'].items() if isinstance(v,dict)}, l['launcher_wall_s'])" &&
$B --dtype f32x3 --steps 100 --warmup 2 > $OUT/pool_100moves_x3_line.json 2> $OUT/pool_100moves.err &&
python3 -c "import json;l=json.load(open('$OUT/pool_100moves_x3_line.json'));print('100 moves', l['value'], l['extra']['tree_pool'], l['extra']['arena_high_water_slots'], l['extra']['truncated_tree_blocks'], l['selfplay_games'])" &&
$B --board 19 --sims 1600 --filters 256 --blocks 40 --games 1024 --steps 2 --warmup 1 --dtype f16r > $OUT/c5_shape_line.json 2> $OUT/c5_shape.err &&
python3 -c "import json;l=json.load(open('$OUT/c5_shape_line.json'));print('C5 shape', l['value'], l['roofline']['frac'], l['extra']['tree_pool'], l['extra']['arena_high_water_slots'], l['extra']['tree_errors'])"
