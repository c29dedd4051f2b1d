#!/bin/bash
# ON THE GPU BOX (one gpurun call): the whole GPU suite, then the driver's bench command; steps joined by &&.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/${1:-run}
mkdir -p $OUT
python -m pytest tests -m gpu -x -q -s --durations=8 2>&1 | grep -v "Invalid move" | tee $OUT/gpu_tests.log | tail -n 14 &&
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err &&
python3 -c "import json;l=json.load(open('$OUT/bench_line.json'));print('driver line', l['value'], l['roofline']['frac'], l['roofline']['traffic'], l['roofline_tree']['tree_ms_per_wave'], l['extra']['net_tflops_end_to_end'], {k:(v.get('value'),v.get('net_tflops_end_to_end'),v.get('leg_wall_s')) for k,v in l['secondary'].items() if isinstance(v,dict)}, l['launcher_wall_s'])"
