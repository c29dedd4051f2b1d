#!/bin/bash
# ON THE GPU BOX: f32 attention core after the DPP / load-placement changes: net parity tests, MainNetwork f32 line + kernel stats
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -o pipefail
mkdir -p gpurun_out/prof_r3tf; RAW=/tmp/prof_r3tf_raw; rm -rf $RAW; mkdir -p $RAW
timeout -k 10 600 python -m pytest tests/test_gpu_net.py tests/test_gpu_split_precision.py -x -q > gpurun_out/r3_att_f32_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r3_att_f32_tests.log; [ $rc -eq 0 ] || exit $rc
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- python3 bench.py --network transgo --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/prof_r3tf/bench_under_rocprof.log 2>&1 || { echo "trace failed"; exit 1; }
cp $(find $RAW/trace -name "*kernel_stats.csv" | head -1) gpurun_out/prof_r3tf/kernel_stats.csv
python3 bench.py --network transgo --steps 10 --warmup 3 --no-cpu-baseline 2> gpurun_out/prof_r3tf/line.err | grep "^{" > gpurun_out/prof_r3tf/line.json
cut -c1-160 gpurun_out/prof_r3tf/line.json
grep "k_attention\|192, " gpurun_out/prof_r3tf/kernel_stats.csv | cut -c1-60,150-260
