#!/bin/bash
# ON THE GPU BOX: SQ counters of the tree kernels on C1-net searches (4096 boards) -> gpurun_out/prof_tree_sq/
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_tree_sq
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY --output-format csv -d $OUT/sq -o sq -- python3 bench.py --sims 64 --filters 32 --blocks 2 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq2 -o sq -- python3 bench.py --sims 64 --filters 32 --blocks 2 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/sq2.log 2>&1
python3 scripts/pmc_summary.py $OUT/sq k_ > $OUT/summary.txt
python3 scripts/pmc_summary.py $OUT/sq2 k_ >> $OUT/summary.txt
