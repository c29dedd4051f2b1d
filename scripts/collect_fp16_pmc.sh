#!/bin/bash
# ON THE GPU BOX: PMC passes for the fp16 conv kernel at F=256 (8192-leaf launches, 9x9), f16 and f16r.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r2_f16
mkdir -p $OUT
ARGS="--filters 256 --blocks 4 --games 2048 --sims 32 --steps 1 --warmup 1 --stagger 0 --no-cpu-baseline"
for D in f16 f16r; do
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/sq_$D -o sq -- python3 bench.py $ARGS --dtype $D > $OUT/sq_$D.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$D -o f -- python3 bench.py $ARGS --dtype $D > $OUT/fetch_$D.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$D -o w -- python3 bench.py $ARGS --dtype $D > $OUT/write_$D.log 2>&1
  python3 bench.py $ARGS --steps 3 --dtype $D 2>/dev/null | grep "^{" > $OUT/line_$D.json
done
ls $OUT
