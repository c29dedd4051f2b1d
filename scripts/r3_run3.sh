#!/bin/bash
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_baseline_sizes.py -x -q > $O/r3_t5.log 2>&1; echo "pytest rc=$?" >> $O/r3_t5.log; tail -4 $O/r3_t5.log
grep -q "pytest rc=0" $O/r3_t5.log || { echo STOP; exit 1; }
python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/r3_b3_slim.json 2> $O/r3_b3_slim.err || { echo STOP bench; exit 1; }
TG_STEM_FULL=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/r3_b3_full.json 2> $O/r3_b3_full.err || { echo STOP bench2; exit 1; }
python - <<'PY'
import json
for n in ("r3_b3_slim", "r3_b3_full"):
    d = json.load(open(f"gpurun_out/{n}.json"))
    print(n, d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["extra"]["net_tflops_end_to_end"])
PY
