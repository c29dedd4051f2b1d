#!/bin/bash
# ON THE GPU BOX: evidence for the three-product split-precision build -- whole GPU suite, tower / MainNetwork / C4-shape f32x3 lines,
# rocprofv3 kernel stats and PMC passes of the tower line (raw data stays in /tmp, summaries travel back)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -o pipefail
OUT=gpurun_out/prof_x3; RAW=/tmp/prof_x3_raw; rm -rf $RAW; mkdir -p $OUT $RAW
say() { echo "[$(date +%T)] $*"; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1
rc=$?; echo "pytest rc=$rc" >> $OUT/gpu_tests.log; tail -3 $OUT/gpu_tests.log; [ $rc -eq 0 ] || exit $rc
X3="--dtype f32x3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/x3_trace -o trace -- python3 bench.py --steps 6 --warmup 2 $X3 > $OUT/x3_bench_under_rocprof.log 2>&1 || { say "x3 trace failed"; exit 1; }
cp $(find $RAW/x3_trace -name "*kernel_stats.csv" | head -1) $OUT/x3_kernel_stats.csv
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $RAW/x3_sq -o sq -- python3 bench.py --steps 1 --warmup 1 $X3 > $RAW/x3_sq.log 2>&1 || { say "x3 sq failed"; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $RAW/x3_fetch -o f -- python3 bench.py --steps 1 --warmup 1 $X3 > $RAW/x3_fetch.log 2>&1 || { say "x3 fetch failed"; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $RAW/x3_write -o w -- python3 bench.py --steps 1 --warmup 1 $X3 > $RAW/x3_write.log 2>&1 || { say "x3 write failed"; exit 1; }
python3 bench.py --steps 20 --warmup 5 $X3 2>/dev/null | grep "^{" > $RAW/line_x3_c2.json
cp $RAW/line_x3_c2.json $OUT/line_x3_c2.json
python3 scripts/pmc_x3_json.py $RAW $OUT/pmc_f32x3.json > $OUT/pmc_f32x3.txt 2>&1; tail -30 $OUT/pmc_f32x3.txt | head -40
say tower done
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/tx_trace -o trace -- python3 bench.py --network transgo --steps 6 --warmup 2 $X3 > $OUT/tx_bench_under_rocprof.log 2>&1 || { say "transgo trace failed"; exit 1; }
cp $(find $RAW/tx_trace -name "*kernel_stats.csv" | head -1) $OUT/tx_kernel_stats.csv
python3 bench.py --network transgo --steps 10 --warmup 3 $X3 2>/dev/null | grep "^{" > $OUT/line_tx3.json
python3 bench.py --board 19 --sims 800 --blocks 20 --filters 256 --games 256 --steps 4 --warmup 1 $X3 2>/dev/null | grep "^{" > $OUT/line_c4.json
python3 -c "
import json
for f in ('line_x3_c2','line_tx3','line_c4'):
    d=json.loads(open('$OUT/%s.json'%f).read().strip().splitlines()[-1]); e=d['extra']
    print(f, d['value'], d['games_per_hour'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], e.get('net_tflops_end_to_end'), 'hw', e['arena_high_water_slots'], 'trunc', e['truncated_tree_blocks'], 'err', e['tree_errors'])"
