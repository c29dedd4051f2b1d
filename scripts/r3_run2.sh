#!/bin/bash
# round-3 measurement batch 2 (all device helpers inlined): GPU suite, default line, 19x19 tree stage ballot vs LDS packing, f32x3 at C4's shape, k_collect phase stamps
O=gpurun_out
fail() { echo "STOP: $1"; exit 1; }
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/r3_t4.log 2>&1; echo "pytest rc=$?" >> $O/r3_t4.log; tail -4 $O/r3_t4.log
grep -q "pytest rc=0" $O/r3_t4.log || fail "gpu suite"
python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/r3_b2.json 2> $O/r3_b2.err || fail "bench default"
T19="--board 19 --sims 200 --filters 128 --blocks 2 --games 1024 --steps 6 --warmup 2 --no-cpu-baseline --dtype f16r"
python bench.py $T19 > $O/r3_tree19_ballot.json 2> $O/r3_tree19_ballot.err || fail "T19 ballot"
C4="--board 19 --sims 800 --filters 256 --blocks 20 --games 256 --steps 2 --warmup 1 --no-cpu-baseline"
python bench.py $C4 --dtype f32x3 > $O/r3_b_f32x3_c4.json 2> $O/r3_b_f32x3_c4.err || fail "C4 f32x3"
cp transgo_amd/libtransgo_hip.so /tmp/lib_keep.so
cp build/libtransgo_hip_lds19.so transgo_amd/libtransgo_hip.so
python bench.py $T19 > $O/r3_tree19_lds.json 2> $O/r3_tree19_lds.err || fail "T19 lds"
cp build/libtransgo_hip_stamp.so transgo_amd/libtransgo_hip.so
python scripts/stamp_tree.py > $O/r3_stamp_tree.txt 2>&1 || fail "stamps"
cp /tmp/lib_keep.so transgo_amd/libtransgo_hip.so
python - <<'PY'
import json
for n in ("r3_b2", "r3_tree19_ballot", "r3_tree19_lds", "r3_b_f32x3_c4"):
    try:
        d = json.load(open(f"gpurun_out/{n}.json"))
        print(n, d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline_tree"]["tree_ms_per_wave"], d["extra"]["net_tflops_end_to_end"])
    except Exception as e:
        print(n, "failed", e)
PY
cat $O/r3_stamp_tree.txt
