#!/bin/bash
# One gpurun call: the tree-facing GPU tests on the chunk pool, then bench lines that show the pool's fill (roomy pool, default pool,
# a 19x19 run).  Steps joined by &&.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r4
mkdir -p $OUT
B="python3 bench.py --no-launcher --no-cpu-baseline"
python -m pytest tests/test_gpu_search.py tests/test_gpu_selfplay.py tests/test_gpu_records.py tests/test_gpu_baseline_sizes.py tests/test_gpu_edges.py tests/test_gpu_rules.py tests/test_gpu_split_precision.py -x -q -s 2>&1 | grep -v "Invalid move" | tee $OUT/pool_tests.log | tail -n 25 &&
$B --steps 20 --warmup 5 --pool-slots 155904 > $OUT/pool_roomy_line.json 2> $OUT/pool_roomy.err &&
python3 -c "import json;l=json.load(open('$OUT/pool_roomy_line.json'));print('roomy', l['value'], l['extra']['tree_pool'], l['extra']['arena_high_water_slots'], l['roofline_tree']['tree_ms_per_wave'], l['selfplay_games']['dropped_arena_overflow'])" &&
$B --steps 20 --warmup 5 > $OUT/pool_default_line.json 2> $OUT/pool_default.err &&
python3 -c "import json;l=json.load(open('$OUT/pool_default_line.json'));print('default', l['value'], l['extra']['tree_pool'], l['extra']['arena_high_water_slots'], l['roofline_tree']['tree_ms_per_wave'], l['selfplay_games']['dropped_arena_overflow'], l['extra']['truncated_tree_blocks'])" &&
$B --board 19 --sims 200 --filters 128 --blocks 2 --games 1024 --steps 30 --warmup 2 --dtype f16r > $OUT/pool_19_line.json 2> $OUT/pool_19.err &&
python3 -c "import json;l=json.load(open('$OUT/pool_19_line.json'));print('19x19', l['value'], l['extra']['tree_pool'], l['extra']['arena_high_water_slots'], l['extra']['tree_errors'], l['selfplay_games']['dropped_arena_overflow'])" &&
bash scripts/ab_lines.sh att_x3 "--network transgo --dtype f32x3 --steps 5 --warmup 2" notouch noperm
