#!/bin/bash
# One gpurun call: pool tests after the fetch-add fix, the default line, same-box A/B of the WU-UCT counter form on a tree-heavy
# workload, and LAST the one confirming run of the round-3 faulting 19x19 build compiled without interprocedural register allocation.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r4
mkdir -p $OUT
B="python3 bench.py --no-launcher --no-cpu-baseline"
python -m pytest tests/test_gpu_search.py tests/test_gpu_edges.py -x -q 2>&1 | grep -v "Invalid move" | tee $OUT/pool_tests2.log | tail -n 6 &&
$B --steps 20 --warmup 5 > $OUT/pool_default_line2.json 2> $OUT/pool_default2.err &&
python3 -c "import json;l=json.load(open('$OUT/pool_default_line2.json'));print('default', l['value'], l['roofline']['achieved'], l['extra']['tree_pool'], l['roofline_tree']['tree_ms_per_wave'], l['selfplay_games']['dropped_arena_overflow'], l['extra']['truncated_tree_blocks'])" &&
bash scripts/ab_lines.sh tree_rmw "--filters 32 --blocks 2 --steps 8 --warmup 2" rmw &&
bash scripts/ab_lines.sh x3_rmw "--dtype f32x3 --steps 8 --warmup 2" rmw &&
( cd build/r3tree && echo "== round-3 tree, k_play<19> -> encode_bits_call<19> out of line, -mllvm -enable-ipra=false ==" &&
  timeout -k 10 300 python3 bench.py --board 19 --sims 200 --filters 128 --blocks 2 --games 1024 --steps 3 --warmup 1 --no-cpu-baseline --dtype f16r > ../../$OUT/fault19_noipra.json 2> ../../$OUT/fault19_noipra.err; echo "fault19 no-IPRA run: rc=$?" | tee ../../$OUT/fault19_noipra.rc; tail -c 600 ../../$OUT/fault19_noipra.err; head -c 300 ../../$OUT/fault19_noipra.json )
