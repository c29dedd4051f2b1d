#!/bin/bash
# ON THE GPU BOX: refresh the default line's PMC traffic + kernel stats after a conv-kernel change (summaries only travel back)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r3; RAW=/tmp/prof_r3_raw2
rm -rf $RAW; mkdir -p $OUT $RAW
timeout -k 10 600 python3 -m pytest tests/test_gpu_net.py tests/test_gpu_baseline_sizes.py -x -q > $OUT/net_tests.log 2>&1; echo "pytest rc=$?" >> $OUT/net_tests.log; tail -2 $OUT/net_tests.log
grep -q "pytest rc=0" $OUT/net_tests.log || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1 || exit 1
cp $(find $RAW/trace -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
python3 scripts/fullbatch_avg.py $(find $RAW/trace -name "*kernel_trace.csv" | head -1) $OUT/bench_conv_fullbatch_avg.json > /dev/null
grep "^{" $OUT/bench_under_rocprof.log > $OUT/bench_line_under_rocprof.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $RAW/fetch -o fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $RAW/bench_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $RAW/write -o write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $RAW/bench_write.log 2>&1 || exit 1
python3 scripts/pmc_json.py $RAW $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1
cat $OUT/pmc_traffic.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/line_default.err | grep "^{" > $OUT/line_default.json
python3 -c "
import json; d=json.load(open('$OUT/line_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['extra']['net_tflops_end_to_end'])"
