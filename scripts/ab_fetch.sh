#!/bin/bash
# FETCH_SIZE of the conv launches for two builds (ON THE GPU BOX): base library vs build/libtransgo_hip_<tag>.so
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp transgo_amd/libtransgo_hip.so /tmp/lib_keep.so
for tag in base "$@"; do
  if [ $tag = base ]; then cp /tmp/lib_keep.so transgo_amd/libtransgo_hip.so; else cp build/libtransgo_hip_$tag.so transgo_amd/libtransgo_hip.so; fi
  rm -rf /tmp/fe_$tag
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/fe_$tag -o f -- python3 scripts/time_net.py 16384 128 6 9 f32 > /tmp/fe_$tag.log 2>&1
  python3 - $tag <<'PY'
import csv, glob, sys, collections, statistics
tag = sys.argv[1]
per = collections.defaultdict(list)
for f in glob.glob(f"/tmp/fe_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "k_conv3x3_sg" in r["Kernel_Name"]:
            k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
            per[k].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
for k, v in sorted(per.items()):
    g = max(x[0] for x in v)
    full = [x[1] for x in v if x[0] >= 0.95 * g]
    print(tag, k, len(full), "launches: FETCH_SIZE x2 =", round(2 * statistics.median(full) / 1024, 1), "MB per launch (rows x 512 B =", round(g / 256 * 192 * 512 / 1e6 if ", 3" in k else g / 256 * 128 * 512 / 1e6, 1), "MB read once)")
PY
done
cp /tmp/lib_keep.so transgo_amd/libtransgo_hip.so
