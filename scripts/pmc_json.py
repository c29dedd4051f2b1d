#!/usr/bin/env python3
"""rocprofv3 --pmc counter_collection CSVs -> per-kernel JSON summary for profiles/ (medians over the launches of the LAST bench
step: the tail of each kernel's dispatch sequence, so the staggering prelude's small searches are left out).
usage: pmc_json.py <fetch_dir> <write_dir> <launches_per_step_conv> <waves_per_step> [out.json]"""
import collections, csv, glob, json, statistics, sys


def load(d, counter):
    per = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
            per[k].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"]),
                           int(r["Grid_Size"])))
    return {k: sorted(v) for k, v in per.items()}


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    n_conv, n_wave = int(sys.argv[3]), int(sys.argv[4])
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --steps 1 --warmup 1 "
                     "--no-cpu-baseline`; medians over the launches of the last (timed) step",
           "correction": "FETCH_SIZE doubled (gfx950 tallies 128-B requests as 64 B: MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact; "
                         "counter unit KB", "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        tail = n_conv if "k_conv3x3_sg" in k or "k_conv3x3_h2" in k else n_wave
        f = [x for x in fetch.get(k, [])][-tail:]
        w = [x for x in write.get(k, [])][-tail:]
        if not f or not w:
            continue
        fk, wk = statistics.median(x[1] for x in f), statistics.median(x[1] for x in w)
        out["kernels"][k] = {"launches_summarised": len(f), "grid_size_median": statistics.median(x[3] for x in f),
                             "FETCH_SIZE_KB_raw_median": round(fk, 1), "WRITE_SIZE_KB_median": round(wk, 1),
                             "hbm_bytes_per_launch": round((2 * fk + wk) * 1024.0, 1),
                             "duration_us_median": round(statistics.median(x[2] for x in f) / 1e3, 2)}
    s = json.dumps(out, indent=1)
    if len(sys.argv) > 5:
        open(sys.argv[5], "w").write(s + "\n")
    else:
        print(s)


if __name__ == "__main__":
    main()
