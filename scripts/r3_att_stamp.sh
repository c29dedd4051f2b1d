#!/bin/bash
# ON THE GPU BOX: stamp build of the fused attention kernel (copied over the product library in this scratch copy only)
cp build/libtransgo_hip_astamp.so transgo_amd/libtransgo_hip.so && timeout -k 10 300 python scripts/stamp_att.py 16384 > gpurun_out/r3_att_stamps.txt 2>&1; rc=$?
cat gpurun_out/r3_att_stamps.txt; exit $rc
