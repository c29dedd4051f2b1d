#!/bin/bash
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_baseline_sizes.py tests/test_gpu_split_precision.py -x -q -s > $O/r3_t7.log 2>&1; echo "pytest rc=$?" >> $O/r3_t7.log; grep -E "head GEMM|passed|failed|Error" $O/r3_t7.log | tail -12
grep -q "pytest rc=0" $O/r3_t7.log || { echo STOP; exit 1; }
python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/r3_b4_gemm.json 2> $O/r3_b4_gemm.err || { echo STOP bench; exit 1; }
TG_HEAD_GEMM=0 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/r3_b4_old.json 2> $O/r3_b4_old.err || { echo STOP bench2; exit 1; }
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --dtype f32x3 > $O/r3_b4_x3.json 2> $O/r3_b4_x3.err || { echo STOP bench3; exit 1; }
python - <<'PY'
import json
for n in ("r3_b4_gemm", "r3_b4_old", "r3_b4_x3"):
    d = json.load(open(f"gpurun_out/{n}.json"))
    print(n, d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["extra"]["net_tflops_end_to_end"])
PY
