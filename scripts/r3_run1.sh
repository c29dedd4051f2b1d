#!/bin/bash
# round-3 measurement batch 1: split-precision parity + lines, 19x19 tree stage with ballot-packed vs LDS-packed planes
set -o pipefail
O=gpurun_out
python -m pytest tests/test_gpu_split_precision.py -x -q > $O/r3_t3.log 2>&1; echo "pytest rc=$?" >> $O/r3_t3.log; tail -12 $O/r3_t3.log
python bench.py --dtype f32x3 --steps 6 --warmup 2 --no-cpu-baseline > $O/r3_b_f32x3_c2.json 2> $O/r3_b_f32x3_c2.err; tail -c 300 $O/r3_b_f32x3_c2.json
C4="--board 19 --sims 800 --filters 256 --blocks 20 --games 256 --steps 2 --warmup 1 --no-cpu-baseline"
python bench.py $C4 --dtype f32x3 > $O/r3_b_f32x3_c4.json 2> $O/r3_b_f32x3_c4.err; tail -c 300 $O/r3_b_f32x3_c4.json
# 19x19 tree stage A/B on a light net (tree share visible): ballot form (shipped) then the LDS form
T19="--board 19 --sims 200 --filters 128 --blocks 2 --games 1024 --steps 3 --warmup 1 --no-cpu-baseline --dtype f16r"
python bench.py $T19 > $O/r3_tree19_ballot.json 2> $O/r3_tree19_ballot.err
cp transgo_amd/libtransgo_hip.so /tmp/lib_keep.so && cp build/libtransgo_hip_lds19.so transgo_amd/libtransgo_hip.so
python bench.py $T19 > $O/r3_tree19_lds.json 2> $O/r3_tree19_lds.err
cp /tmp/lib_keep.so transgo_amd/libtransgo_hip.so
python - <<'PY'
import json
for n in ("ballot", "lds"):
    try:
        d = json.load(open(f"gpurun_out/r3_tree19_{n}.json"))
        print(n, d["value"], d["roofline_tree"]["tree_ms_per_wave"], d["ms_per_step"])
    except Exception as e:
        print(n, "failed", e)
PY
