#!/usr/bin/env python3
"""rocprofv3 per-dispatch kernel trace -> average duration of the F->F conv's FULL-BATCH launches (the --stats averages also
contain the root-evaluation launches of a few dozen rows).  usage: fullbatch_avg.py <trace_kernel_trace.csv> <out.json>"""
import collections, csv, json, sys

per = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
    if k.startswith("k_conv3x3_sg") or k.startswith("k_conv3x3_h2"):
        per[k].append((int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
out = {"source": "the run of the kernel-stats file beside this one (rocprofv3 --kernel-trace --stats on `python3 bench.py --no-launcher --steps 20 --warmup 5 "
                 "--no-cpu-baseline`), per-dispatch trace: launches of the F->F conv whose grid is within 5 % of the largest (full leaf "
                 "batches; the --stats averages also contain the root-evaluation launches of a few dozen rows)", "kernels": {}}
for k, v in sorted(per.items()):
    g = max(x[0] for x in v)
    full = [d for gs, d in v if gs >= 0.95 * g]
    out["kernels"][k] = {"launches": len(v), "full_batch_launches": len(full), "avg_us_all": round(sum(d for _, d in v) / len(v) / 1e3, 1),
                         "avg_us_full_batch": round(sum(full) / len(full) / 1e3, 1)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
