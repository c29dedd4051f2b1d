#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_split_precision.py tests/test_gpu_search.py -x -q -s -k "mainnetwork or peaked" > gpurun_out/r3_last_tests.log 2>&1
rc=$?; grep -a "boards: max\|high-water\|passed\|failed\|Error\|assert" gpurun_out/r3_last_tests.log | tail -12; exit $rc
