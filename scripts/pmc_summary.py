"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (development aid): mean counter value per launch."""
import collections, csv, glob, sys
pat = sys.argv[2] if len(sys.argv) > 2 else "k_conv3x3"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[1 if r["Kernel_Name"].startswith("void (") else 0][:0] or r["Kernel_Name"]
            k = k.replace("void (anonymous namespace)::", "").split("(")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[k]["_dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, d in acc.items():
    n = len(next(iter(d.values())))
    print(k, f"[{n} samples]")
    for c, v in sorted(d.items()):
        v = sorted(v)
        print(f"    {c:32s} mean {sum(v)/len(v):16.1f}   median {v[len(v)//2]:16.1f}")
