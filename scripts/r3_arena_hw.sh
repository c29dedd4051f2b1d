#!/bin/bash
# ON THE GPU BOX: true arena high-water marks (per-slot maximum over the run) of the default line and of the MainNetwork lines
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_search.py -x -q -s -k "peaked or arena" 2>&1 | grep -a "high-water\|passed\|failed" | tee gpurun_out/r3_arena_hw.txt
pr() { python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); e=d['extra']
print(sys.argv[2], 'sims/s', d['value'], 'steps', d['steps'], 'high-water', e['arena_high_water_slots'], 'of', e['arena_slots_per_half'], 'truncated blocks', e['truncated_tree_blocks'], 'errors', e['tree_errors'])" $1 "$2" | tee -a gpurun_out/r3_arena_hw.txt; }
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/hw_default.json 2>/dev/null || exit 1; pr gpurun_out/hw_default.json "default (6x128 tower, f32)"
timeout -k 10 500 python bench.py --network transgo --dtype f32x3 --steps 40 --warmup 2 --no-cpu-baseline > gpurun_out/hw_tx3.json 2>/dev/null || exit 1; pr gpurun_out/hw_tx3.json "MainNetwork f32x3, 40 moves"
timeout -k 10 500 python bench.py --network transgo --dtype f32x3 --steps 40 --warmup 2 --no-cpu-baseline --arena-slots 244608 > gpurun_out/hw_tx3_big.json 2>/dev/null || exit 1; pr gpurun_out/hw_tx3_big.json "MainNetwork f32x3, 40 moves, arena x2"
