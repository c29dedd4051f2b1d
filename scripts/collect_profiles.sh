#!/bin/bash
# Run ON THE GPU BOX (gpurun): rocprofv3 kernel trace of the driver's bench command + separate PMC passes (FETCH_SIZE, WRITE_SIZE)
# as MI355X_MICROARCH.md prescribes (counters in their own runs, --kernel-trace only).  Output under gpurun_out/prof_r2/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r2
mkdir -p $OUT
EXTRA="$@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $EXTRA > $OUT/bench_trace.log 2>&1
echo trace done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/bench_fetch.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/bench_write.log 2>&1
echo write done
ls -R $OUT | head -30
