#!/bin/bash
# ON THE GPU BOX: same-box A/B of library build variants (build/libtransgo_hip_<v>.so copied over the product library in this scratch
# copy only; "prod" = the product build), bench lines interleaved, two repetitions.
#   scripts/ab_lines.sh <tag> "<bench.py arguments>" <variant> [<variant> ...]
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
tag=$1; args=$2; shift 2
mkdir -p gpurun_out/r4
cp transgo_amd/libtransgo_hip.so /tmp/prod.so
for rep in 1 2; do
  for v in prod "$@"; do
    if [ $v = prod ]; then cp /tmp/prod.so transgo_amd/libtransgo_hip.so; else cp build/libtransgo_hip_$v.so transgo_amd/libtransgo_hip.so; fi
    timeout -k 10 300 python3 bench.py --no-launcher --no-cpu-baseline $args 2> gpurun_out/r4/ab_$tag.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline_tree']['tree_ms_per_wave'])" | tee -a gpurun_out/r4/ab_$tag.txt || { cp /tmp/prod.so transgo_amd/libtransgo_hip.so; exit 1; }
  done
done
cp /tmp/prod.so transgo_amd/libtransgo_hip.so
