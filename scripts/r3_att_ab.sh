#!/bin/bash
# ON THE GPU BOX: A/B of fused-attention build variants (build/libtransgo_hip_<v>.so copied over the product library in this scratch
# copy only), MainNetwork f32x3 lines, two repetitions interleaved.  usage: r3_att_ab.sh <variant> [<variant> ...]
set -o pipefail
mkdir -p gpurun_out
cp transgo_amd/libtransgo_hip.so /tmp/prod.so
for rep in 1 2; do
  for v in prod "$@"; do
    if [ $v = prod ]; then cp /tmp/prod.so transgo_amd/libtransgo_hip.so; else cp build/libtransgo_hip_$v.so transgo_amd/libtransgo_hip.so; fi
    timeout -k 10 300 python bench.py --network transgo --dtype f32x3 --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/r3_att_ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r3_att_ab.txt || exit 1
  done
done
cp /tmp/prod.so transgo_amd/libtransgo_hip.so
