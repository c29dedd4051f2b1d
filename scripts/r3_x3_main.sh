#!/bin/bash
# MainNetwork under split precision: parity tests, then bench lines (f32x3 and f32 on the same box)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_split_precision.py tests/test_gpu_net.py -x -q -s > gpurun_out/r3_x3_main_tests.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r3_x3_main_tests.log; tail -5 gpurun_out/r3_x3_main_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --network transgo --dtype f32x3 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3_transgo_x3_line.json 2> gpurun_out/r3_transgo_x3.err || { tail -20 gpurun_out/r3_transgo_x3.err; exit 1; }
cut -c1-400 gpurun_out/r3_transgo_x3_line.json
timeout -k 10 300 python bench.py --network transgo --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3_transgo_f32_line_b.json 2> gpurun_out/r3_transgo_f32_b.err || { tail -20 gpurun_out/r3_transgo_f32_b.err; exit 1; }
cut -c1-400 gpurun_out/r3_transgo_f32_line_b.json
