#!/bin/bash
# Second half of the round-3 evidence (run ON THE GPU BOX): the long 19x19 run, C5's shape on one GPU, the whole GPU suite.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r3
mkdir -p $OUT
say() { echo "[$(date +%T)] $*"; }
# long 19x19 run of the shipped build (ballot-packed planes, no out-of-line device calls): 1024 boards x 120 moves x 200 sims
python3 bench.py --board 19 --sims 200 --filters 128 --blocks 2 --games 1024 --steps 120 --warmup 2 --no-cpu-baseline --dtype f16r 2> $OUT/soak19.err | grep "^{" > $OUT/soak19_line.json
say soak19 done
C5="--board 19 --sims 1600 --filters 256 --blocks 40 --games 1024 --steps 2 --warmup 1 --no-cpu-baseline"
python3 bench.py $C5 --dtype f16r 2> $OUT/c5_f16r.err | grep "^{" > $OUT/line_c5_f16r.json
say c5 done
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -s > $OUT/gpu_tests_full.log 2>&1; echo "pytest rc=$?" >> $OUT/gpu_tests_full.log
grep -E "f32x3|head GEMM|passed|failed|MainNetwork through" $OUT/gpu_tests_full.log > $OUT/gpu_tests_summary.txt
du -sh $OUT; ls -la $OUT
