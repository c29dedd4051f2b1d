#!/bin/bash
# ON THE GPU BOX: the whole GPU suite and smoke() on the final tree
set -o pipefail
mkdir -p gpurun_out/final
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/final/gpu_tests.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/final/gpu_tests.log; tail -3 gpurun_out/final/gpu_tests.log; [ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/final/gpu_tests.log
