#!/bin/bash
# ON THE GPU BOX: final tree -- full GPU suite, the default line (driver's command), the MainNetwork split-precision line, and a true
# arena high-water for C4's shape (19x19, 800 sims, 20x256, 256 boards, f32x3, a few moves)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -o pipefail
mkdir -p gpurun_out/final
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/final/gpu_tests.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/final/gpu_tests.log; tail -3 gpurun_out/final/gpu_tests.log; [ $rc -eq 0 ] || exit $rc
python3 bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/final/line_default.err | grep "^{" > gpurun_out/final/line_default.json || exit 1
python3 bench.py --network transgo --dtype f32x3 --steps 10 --warmup 3 --no-cpu-baseline 2> gpurun_out/final/line_tx3.err | grep "^{" > gpurun_out/final/line_tx3.json || exit 1
python3 bench.py --board 19 --sims 800 --blocks 20 --filters 256 --games 256 --dtype f32x3 --steps 4 --warmup 1 --no-cpu-baseline 2> gpurun_out/final/line_c4.err | grep "^{" > gpurun_out/final/line_c4.json || exit 1
python3 -c "
import json
for f in ('line_default','line_tx3','line_c4'):
    d=json.loads(open('gpurun_out/final/%s.json'%f).read().strip().splitlines()[-1]); e=d['extra']
    print(f, d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], 'hw', e['arena_high_water_slots'], 'of', e['arena_slots_per_half'], 'trunc', e['truncated_tree_blocks'], 'err', e['tree_errors'])"
