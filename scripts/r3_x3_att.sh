#!/bin/bash
# fused split-precision attention: parity tests, then MainNetwork lines fused vs the two-kernel form (same box)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_split_precision.py -x -q -s -k "mainnetwork" > gpurun_out/r3_x3_att_tests.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r3_x3_att_tests.log; tail -6 gpurun_out/r3_x3_att_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --network transgo --dtype f32x3 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3_x3_att_fused.json 2> gpurun_out/r3_x3_att_fused.err || { tail -20 gpurun_out/r3_x3_att_fused.err; exit 1; }
cut -c1-200 gpurun_out/r3_x3_att_fused.json
TG_ATT_X3=0 timeout -k 10 300 python bench.py --network transgo --dtype f32x3 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3_x3_att_pair.json 2> gpurun_out/r3_x3_att_pair.err || { tail -20 gpurun_out/r3_x3_att_pair.err; exit 1; }
cut -c1-200 gpurun_out/r3_x3_att_pair.json
