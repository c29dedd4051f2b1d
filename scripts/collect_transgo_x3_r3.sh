#!/bin/bash
# ON THE GPU BOX: the reference's shipped MainNetwork (--network transgo) under split precision: line + rocprofv3 kernel stats (summaries only travel back)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r3tx; RAW=/tmp/prof_r3tx_raw
rm -rf $RAW; mkdir -p $OUT $RAW
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- python3 bench.py --network transgo --dtype f32x3 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1 || { echo "trace failed"; exit 1; }
cp $(find $RAW/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 bench.py --network transgo --dtype f32x3 --steps 10 --warmup 3 --no-cpu-baseline 2> $OUT/line.err | grep "^{" > $OUT/line.json
ls -la $OUT
