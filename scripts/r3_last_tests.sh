#!/bin/bash
# ON THE GPU BOX: network parity tests of the final tree
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_split_precision.py tests/test_gpu_selfplay.py tests/test_gpu_net.py -x -q -s > gpurun_out/r3_last_tests.log 2>&1
rc=$?; grep -a "MainNetwork\|passed\|failed\|Error\|assert" gpurun_out/r3_last_tests.log | tail -12; exit $rc
