#!/bin/bash
# gpurun call B of the round's final evidence: MainNetwork kernel stats (f32, f32x3), PMC of the fused attention block, SQ counters of
# the tree kernels, and BASELINE configs[3]'s shape (19x19, 800 sims, 20x256, 1024 boards, exact f32) on the pool.
cd "${GRAFT_REPO_ROOT:-.}"
export PYTHONUNBUFFERED=1
bash scripts/collect_profiles.sh r4 transgo att tree &&
python3 bench.py --no-launcher --no-cpu-baseline --board 19 --sims 800 --filters 256 --blocks 20 --games 1024 --steps 2 --warmup 1 > gpurun_out/prof_r4/c4_shape_f32_line.json 2> gpurun_out/prof_r4/c4_shape.err &&
python3 -c "import json;l=json.load(open('gpurun_out/prof_r4/c4_shape_f32_line.json'));print('C4 shape', l['value'], l['roofline']['achieved'], l['roofline']['frac'], l['extra']['tree_pool'], l['extra']['net_tflops_end_to_end'])"
