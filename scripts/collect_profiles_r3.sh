#!/bin/bash
# Run ON THE GPU BOX (gpurun).  Round-3 evidence: rocprofv3 kernel trace of the driver's bench command, PMC passes in their own runs
# (FETCH_SIZE, WRITE_SIZE; SQ counters for the split-precision conv), the split-precision lines, the N = 2 launcher line.
# Raw rocprofv3 output stays on the box (it is tens of MB); only the summaries under gpurun_out/prof_r3/ travel back.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r3
RAW=/tmp/prof_r3_raw
rm -rf $RAW; mkdir -p $OUT $RAW
say() { echo "[$(date +%T)] $*"; }
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1 || { say "trace failed"; exit 1; }
cp $(find $RAW/trace -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
python3 scripts/fullbatch_avg.py $(find $RAW/trace -name "*kernel_trace.csv" | head -1) $OUT/bench_conv_fullbatch_avg.json > /dev/null
grep "^{" $OUT/bench_under_rocprof.log > $OUT/bench_line_under_rocprof.json
say trace done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $RAW/fetch -o fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $RAW/bench_fetch.log 2>&1 || { say "fetch failed"; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $RAW/write -o write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $RAW/bench_write.log 2>&1 || { say "write failed"; exit 1; }
python3 scripts/pmc_json.py $RAW $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1
say default pmc done
X3="--dtype f32x3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/x3_trace -o trace -- python3 bench.py --steps 6 --warmup 2 $X3 > $OUT/x3_bench_under_rocprof.log 2>&1 || { say "x3 trace failed"; exit 1; }
cp $(find $RAW/x3_trace -name "*kernel_stats.csv" | head -1) $OUT/x3_kernel_stats.csv
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $RAW/x3_sq -o sq -- python3 bench.py --steps 1 --warmup 1 $X3 > $RAW/x3_sq.log 2>&1 || { say "x3 sq failed"; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $RAW/x3_fetch -o f -- python3 bench.py --steps 1 --warmup 1 $X3 > $RAW/x3_fetch.log 2>&1 || { say "x3 fetch failed"; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $RAW/x3_write -o w -- python3 bench.py --steps 1 --warmup 1 $X3 > $RAW/x3_write.log 2>&1 || { say "x3 write failed"; exit 1; }
python3 bench.py --steps 20 --warmup 5 $X3 2>/dev/null | grep "^{" > $RAW/line_x3_c2.json
cp $RAW/line_x3_c2.json $OUT/line_x3_c2.json
python3 scripts/pmc_x3_json.py $RAW $OUT/pmc_f32x3.json > $OUT/pmc_f32x3.txt 2>&1
say x3 done
python3 bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/line_default.err | grep "^{" > $OUT/line_default.json
say default line done
TRANSGO_DIST_BACKEND=gloo python3 bench.py --gpus 2 --games 1024 --steps 3 --warmup 1 2> $OUT/line_n2.err | grep "^{" > $OUT/line_n2_gloo.json
say n2 done
du -sh $OUT; ls -la $OUT
