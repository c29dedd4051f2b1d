#!/bin/bash
# A/B of conv-kernel variants built into build/libtransgo_hip_<tag>.so: forward timing at 16384 leaves (scripts/time_net.py), same box
cp transgo_amd/libtransgo_hip.so /tmp/lib_keep.so
for rep in 1 2; do
for tag in base "$@"; do
  if [ $tag = base ]; then cp /tmp/lib_keep.so transgo_amd/libtransgo_hip.so; else cp build/libtransgo_hip_$tag.so transgo_amd/libtransgo_hip.so; fi
  echo -n "$tag: "; python scripts/time_net.py 16384 128 6 9 f32 2>/dev/null | tail -1
  echo -n "$tag (15700 leaves): "; python scripts/time_net.py 15700 128 6 9 f32 2>/dev/null | tail -1
done
done
cp /tmp/lib_keep.so transgo_amd/libtransgo_hip.so
