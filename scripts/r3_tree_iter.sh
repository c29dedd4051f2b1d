#!/bin/bash
# tree-stage iteration: bit-exact parity subset, then the stamped build's phase table, then the tree-stage time on a light net
O=gpurun_out; T=${1:-x}
timeout -k 10 600 python -m pytest tests/test_gpu_rules.py tests/test_gpu_search.py tests/test_gpu_records.py -x -q > $O/r3_tree_${T}_tests.log 2>&1; echo "pytest rc=$?" >> $O/r3_tree_${T}_tests.log; tail -3 $O/r3_tree_${T}_tests.log
grep -q "pytest rc=0" $O/r3_tree_${T}_tests.log || { echo STOP; exit 1; }
python bench.py --filters 32 --blocks 2 --steps 6 --warmup 2 --no-cpu-baseline > $O/r3_tree_${T}_c1net.json 2> $O/r3_tree_${T}_c1net.err || { echo STOP bench; exit 1; }
python -c "
import json; d=json.load(open('$O/r3_tree_${T}_c1net.json')); print('C1-net line', d['value'], d['ms_per_step'], d['roofline_tree'])"
cp transgo_amd/libtransgo_hip.so /tmp/lib_keep.so; cp build/libtransgo_hip_stamp.so transgo_amd/libtransgo_hip.so
python scripts/stamp_tree.py > $O/r3_tree_${T}_stamps.txt 2>&1; cp /tmp/lib_keep.so transgo_amd/libtransgo_hip.so
cat $O/r3_tree_${T}_stamps.txt
