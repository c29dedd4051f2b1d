#!/bin/bash
# A/B of network builds (build/libtransgo_hip_<tag>.so vs the shipped library): scripts/time_net.py lines, same box, two repetitions
# usage: ab_net.sh "<time_net args>" tag [tag ...]     e.g. ab_net.sh "16384 128 6 9 f16r" hx
ARGS="$1"; shift
cp transgo_amd/libtransgo_hip.so /tmp/lib_keep.so
for rep in 1 2; do
for tag in base "$@"; do
  if [ $tag = base ]; then cp /tmp/lib_keep.so transgo_amd/libtransgo_hip.so; else cp build/libtransgo_hip_$tag.so transgo_amd/libtransgo_hip.so; fi
  echo -n "$tag: "; python scripts/time_net.py $ARGS 2>/dev/null | tail -1
done
done
cp /tmp/lib_keep.so transgo_amd/libtransgo_hip.so
