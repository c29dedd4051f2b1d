#!/bin/bash
# Run ON THE GPU BOX (gpurun): regenerates the summaries committed under profiles/ for the current round.
#   scripts/collect_profiles.sh <tag> <section> [<section> ...]      sections: default x3 transgo att tree
# rocprofv3 wraps `python3 bench.py --no-launcher --no-cpu-baseline ...` directly (the profiled process is the rank itself: a process
# whose GPU the profiler has already initialised must not start GPU children).  Counters (--pmc) are collected in their own passes
# with --kernel-trace only.  Raw output stays on the box under /tmp; the summaries land in gpurun_out/prof_<tag>/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
RAW=/tmp/prof_${TAG}_raw
mkdir -p $OUT $RAW
B="python3 bench.py --no-launcher --no-cpu-baseline"
say() { echo "[$(date +%T)] $*"; }
trace() {   # <name> <bench args...>: kernel trace + stats, per-dispatch full-batch averages, the line measured under the profiler
  local n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/${n}_trace -o trace -- $B "$@" > $OUT/${n}_under_rocprof.log 2>&1 || { say "$n trace failed"; return 1; }
  cp $(find $RAW/${n}_trace -name "*kernel_stats.csv" | head -1) $OUT/${n}_kernel_stats.csv
  python3 scripts/fullbatch_avg.py $(find $RAW/${n}_trace -name "*kernel_trace.csv" | head -1) $OUT/${n}_conv_fullbatch_avg.json > /dev/null
  grep "^{" $OUT/${n}_under_rocprof.log > $OUT/${n}_line_under_rocprof.json
  say "$n trace done"
}
pmc() {     # <dir> <counters...> -- <bench args...>
  local d=$1; shift; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rocprofv3 --kernel-trace --pmc "${ctr[@]}" --output-format csv -d $RAW/$d -o pmc -- $B "$@" > $RAW/$d.log 2>&1 || { say "$d failed"; return 1; }
}
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY"
for sec in "$@"; do
  case $sec in
  default)   # BASELINE configs[1], exact f32: the driver's command
    trace bench --steps 20 --warmup 5 || exit 1
    pmc fetch FETCH_SIZE -- --steps 1 --warmup 1 && pmc write WRITE_SIZE -- --steps 1 --warmup 1 || exit 1
    python3 scripts/pmc_json.py $RAW $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1; say "default pmc done" ;;
  x3)        # the same workload under the split-precision network
    trace x3 --dtype f32x3 --steps 6 --warmup 2 || exit 1
    pmc x3_sq $SQ -- --dtype f32x3 --steps 1 --warmup 1 && pmc x3_fetch FETCH_SIZE -- --dtype f32x3 --steps 1 --warmup 1 && pmc x3_write WRITE_SIZE -- --dtype f32x3 --steps 1 --warmup 1 || exit 1
    $B --dtype f32x3 --steps 20 --warmup 5 2>/dev/null | grep "^{" > $RAW/line_x3_c2.json; cp $RAW/line_x3_c2.json $OUT/line_x3_c2.json
    python3 scripts/pmc_x3_json.py $RAW $OUT/pmc_f32x3.json > $OUT/pmc_f32x3.txt 2>&1; say "x3 done" ;;
  transgo)   # the reference's shipped MainNetwork, exact f32 and split precision
    trace transgo --network transgo --steps 4 --warmup 1 || exit 1
    trace transgo_x3 --network transgo --dtype f32x3 --steps 6 --warmup 2 || exit 1 ;;
  att)       # PMC on the fused split-precision attention block
    A="--network transgo --dtype f32x3 --steps 1 --warmup 1"
    mkdir -p $RAW/att
    pmc att/sq $SQ -- $A && pmc att/fetch FETCH_SIZE -- $A && pmc att/write WRITE_SIZE -- $A || exit 1
    python3 scripts/pmc_att_json.py $RAW/att $OUT/pmc_attention_x3.json > $OUT/pmc_attention_x3.txt 2>&1; say "att done" ;;
  tree)      # SQ counters of the tree kernels on C1-net searches (4096 boards)
    T="--sims 64 --filters 32 --blocks 2 --steps 2 --warmup 1"
    pmc tree_sq SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY -- $T || exit 1
    pmc tree_sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY -- $T || exit 1
    python3 scripts/pmc_summary.py $RAW/tree_sq k_ > $OUT/tree_sq_summary.txt; python3 scripts/pmc_summary.py $RAW/tree_sq2 k_ >> $OUT/tree_sq_summary.txt; say "tree done" ;;
  *) say "unknown section $sec"; exit 2 ;;
  esac
done
du -sh $OUT; ls -la $OUT
