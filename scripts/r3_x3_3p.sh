#!/bin/bash
# ON THE GPU BOX: three-product split conv -- parity tests, then tower and MainNetwork f32x3 lines against the four-product build
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_split_precision.py -x -q -s > gpurun_out/r3_x3_3p_tests.log 2>&1
rc=$?; grep -a "f32x3\|passed\|failed\|Error\|assert" gpurun_out/r3_x3_3p_tests.log | tail -12
[ $rc -eq 0 ] || exit $rc
cp transgo_amd/libtransgo_hip.so /tmp/prod.so
for rep in 1 2; do
  for v in prod p4; do
    if [ $v = prod ]; then cp /tmp/prod.so transgo_amd/libtransgo_hip.so; else cp build/libtransgo_hip_$v.so transgo_amd/libtransgo_hip.so; fi
    for net in tower transgo; do
      timeout -k 10 300 python bench.py --network $net --dtype f32x3 --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/r3_x3_3p.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$net', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'])" | tee -a gpurun_out/r3_x3_3p.txt || exit 1
    done
  done
done
cp /tmp/prod.so transgo_amd/libtransgo_hip.so
