#!/bin/bash
# ON THE GPU BOX: parity of the MainNetwork split-precision tests, A/B against build/libtransgo_hip_prev.so, PMC summary of the fused kernel
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_split_precision.py tests/test_gpu_selfplay.py -x -q -s -k "mainnetwork" > gpurun_out/r3_x3_att_tests.log 2>&1
rc=$?; grep -a "MainNetwork\|passed\|failed" gpurun_out/r3_x3_att_tests.log | tail -8
[ $rc -eq 0 ] || exit $rc
rm -f gpurun_out/r3_att_ab.txt
bash scripts/r3_att_ab.sh prev || exit 1
bash scripts/collect_pmc_att_r3.sh
