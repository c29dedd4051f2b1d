#!/bin/bash
# gpurun call D: the final build (batched chunk reservation in k_play, on-demand pops in k_collect): tree / records / bench tests,
# kernel stats of the tree-heavy workload, then the whole suite and the driver's command.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-.}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r4
mkdir -p $OUT
python -m pytest tests/test_gpu_search.py tests/test_gpu_edges.py tests/test_gpu_records.py -x -q 2>&1 | grep -v "Invalid move" | tee $OUT/pool_tests4.log | tail -n 4 &&
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c1_trace -o trace -- python3 bench.py --no-launcher --no-cpu-baseline --sims 64 --filters 32 --blocks 2 --steps 2 --warmup 1 > $OUT/c1net_under_rocprof.log 2>&1 &&
cp $(find /tmp/c1_trace -name "*kernel_stats.csv" | head -1) $OUT/c1net_sims64_kernel_stats.csv && grep -E "k_play|k_collect|k_absorb" $OUT/c1net_sims64_kernel_stats.csv | cut -c1-200 &&
python3 bench.py --no-launcher --no-cpu-baseline --filters 32 --blocks 2 --steps 8 --warmup 2 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C1 net, 400 sims:', d['value'], d['ms_per_step'], d['roofline_tree']['tree_ms_per_wave'], d['extra']['step_phases_ms'])" &&
python -m pytest tests -m gpu -x -q -s --durations=8 2>&1 | grep -v "Invalid move" | tee $OUT/gpu_tests_final2.log | tail -n 14 &&
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line_final2.json 2> $OUT/bench_final2.err &&
python3 -c "import json;l=json.load(open('$OUT/bench_line_final2.json'));print('driver line', l['value'], l['roofline']['frac'], l['roofline']['traffic'], l['roofline_tree']['tree_ms_per_wave'], l['extra']['tree_pool']['high_water_frac'], l['extra']['step_phases_ms'], {k:(v.get('value'),v.get('leg_wall_s')) for k,v in l['secondary'].items() if isinstance(v,dict)}, l['launcher_wall_s'])"
