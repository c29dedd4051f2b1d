#!/bin/bash
# ON THE GPU BOX: full GPU suite on the final tree, then the MainNetwork split-precision line + kernel stats
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -o pipefail
mkdir -p gpurun_out/prof_r3tx; RAW=/tmp/prof_r3tx_raw; rm -rf $RAW; mkdir -p $RAW
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gpu_tests_final.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r3_gpu_tests_final.log; tail -4 gpurun_out/r3_gpu_tests_final.log; [ $rc -eq 0 ] || exit $rc
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- python3 bench.py --network transgo --dtype f32x3 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/prof_r3tx/bench_under_rocprof.log 2>&1 || { echo "trace failed"; exit 1; }
cp $(find $RAW/trace -name "*kernel_stats.csv" | head -1) gpurun_out/prof_r3tx/kernel_stats.csv
python3 bench.py --network transgo --dtype f32x3 --steps 10 --warmup 3 --no-cpu-baseline 2> gpurun_out/prof_r3tx/line.err | grep "^{" > gpurun_out/prof_r3tx/line.json
cut -c1-160 gpurun_out/prof_r3tx/line.json
grep "k_attention" gpurun_out/prof_r3tx/kernel_stats.csv | cut -c1-70,190-290
