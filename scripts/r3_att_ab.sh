#!/bin/bash
# ON THE GPU BOX: fused attention with / without the next-board touch prefetch (two builds, same box), MainNetwork f32x3 lines
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
  for v in touch notouch; do
    if [ $v = notouch ]; then cp transgo_amd/libtransgo_hip.so /tmp/prod.so; cp build/libtransgo_hip_notouch.so transgo_amd/libtransgo_hip.so; fi
    timeout -k 10 300 python bench.py --network transgo --dtype f32x3 --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/r3_att_ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r3_att_ab.txt || exit 1
    if [ $v = notouch ]; then cp /tmp/prod.so transgo_amd/libtransgo_hip.so; fi
  done
done
