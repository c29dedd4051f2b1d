#!/usr/bin/env python3
"""Summarise the split-precision passes of scripts/collect_profiles.sh (gpurun_out/prof_r3/x3_*) into profiles/r3_pmc_f32x3.json:
MFMA pipe busy, wave-cycle waiting share and HBM traffic of k_conv3x3_h2<9,256,128,EPI,false,X2> at BASELINE configs[1]'s shape."""
import collections, csv, glob, json, re, statistics, sys
ROOT = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r3"
OUT = sys.argv[2] if len(sys.argv) > 2 else "profiles/r3_pmc_f32x3.json"
F = 128


def load(d):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_conv3x3_h2ILi9ELi256ELi128ELi(\d)ELb0ELb1", r["Kernel_Name"]) or \
                re.search(r"k_conv3x3_h2<9, 256, 128, (\d), false, true>", r["Kernel_Name"])
            if m:
                per[m.group(1)][r["Counter_Name"]].append((float(r["Counter_Value"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), int(r["Grid_Size"])))
    return per


sq, fe, wr = load(f"{ROOT}/x3_sq"), load(f"{ROOT}/x3_fetch"), load(f"{ROOT}/x3_write")
line = json.loads(open(f"{ROOT}/line_x3_c2.json").read())
out = {"source": "rocprofv3 --kernel-trace --pmc ... (separate passes: SQ counters, FETCH_SIZE, WRITE_SIZE; scripts/collect_profiles.sh) on "
                 "`python3 bench.py --no-launcher --no-cpu-baseline --dtype f32x3 --steps 1 --warmup 1` (9x9, 400 sims, 6x128, 4096 boards); medians over the "
                 "full-batch launches (grid within 5 % of the largest) of k_conv3x3_h2<9,256,128,EPI,false,X2>",
       "correction": "FETCH_SIZE doubled (gfx950: MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact; unit KB.  MFMA pipe busy = "
                     "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs)", "kernels": {}}
for k in sorted(sq):
    gmax = max(x[2] for x in sq[k]["GRBM_GUI_ACTIVE"])
    sel = lambda per, c: [x for x in per[k][c] if x[2] >= 0.95 * gmax]
    med = lambda per, c: statistics.median(x[0] for x in sel(per, c))
    dur = statistics.median(x[1] for x in sel(sq, "GRBM_GUI_ACTIVE"))
    rows = statistics.median(x[2] for x in sel(sq, "GRBM_GUI_ACTIVE"))            # one work-item per row (256-row tiles of 256 threads)
    mf, gui, wc, wa, f, w = med(sq, "SQ_VALU_MFMA_BUSY_CYCLES"), med(sq, "GRBM_GUI_ACTIVE"), med(sq, "SQ_WAVE_CYCLES"), med(sq, "SQ_WAIT_ANY"), med(fe, "FETCH_SIZE"), med(wr, "WRITE_SIZE")
    alg = (8 * F if k == "0" else 16 * F) * rows
    name = {"0": "EPI 0 (conv1: relu -> h hi/lo)", "1": "EPI 1 (conv2: + f32 residual, writes residual stream + next activation hi/lo)"}[k]
    out["kernels"][name] = {"launches_summarised": len(sel(sq, "GRBM_GUI_ACTIVE")), "rows_per_launch_approx": int(rows),
                            "launch_us_median_under_pmc": round(dur / 1e3, 1), "mfma_pipe_busy": round(mf / (gui / 8 * 1024), 3),
                            "clock_ghz_effective": round(gui / 8 / dur, 3), "tflops_effective_under_pmc": round(2 * 9 * F * F * rows / dur / 1e3, 1),
                            "wave_cycles_waiting_share": round(wa / wc, 3), "FETCH_SIZE_KB_raw": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
                            "hbm_bytes_per_launch": round((2 * f + w) * 1024), "algorithmic_bytes_per_launch": int(alg),
                            "traffic_over_algorithmic": round((2 * f + w) * 1024 / alg, 3), "hbm_tb_per_s": round((2 * f + w) * 1024 / dur / 1e3, 2)}
ks = list(out["kernels"].values())
out["hbm_bytes_per_launch_mean"] = sum(v["hbm_bytes_per_launch"] for v in ks) / max(1, len(ks))
out["bench_line_unprofiled"] = {"value_sims_per_s": line["value"], "conv_tflops_effective": line["roofline"]["achieved"], "frac_of_fp16_peak_over_3": line["roofline"]["frac"],
                                "avg_launch_ms": line["roofline"]["avg_launch_ms"], "net_tflops_end_to_end": line["extra"]["net_tflops_end_to_end"]}
import os, sys as _sys
_sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import net_source_hash
out["net_hip_sha16"] = net_source_hash()      # bench.py reports `traffic` only while the network kernels' code is this (comments apart)
json.dump(out, open(OUT, "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
