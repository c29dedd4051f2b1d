#!/usr/bin/env python3
"""Summarise scripts/collect_fp16_pmc.sh's passes (gpurun_out/prof_r2_f16) into profiles/r2_pmc_f16_conv256.json."""
import collections, csv, glob, json, re, statistics, sys
ROOT = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r2_f16"


def load(d):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_conv3x3_h2ILi9ELi256ELi256ELi(\d)ELb(\d)", r["Kernel_Name"])
            if m:
                per[m.group(1)][r["Counter_Name"]].append((float(r["Counter_Value"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), int(r["Grid_Size"])))
    return per


out = {"source": "rocprofv3 --kernel-trace --pmc ... (separate passes: SQ counters, FETCH_SIZE, WRITE_SIZE) on `python3 bench.py --filters 256 "
                 "--blocks 4 --games 2048 --sims 32 --steps 1 --warmup 1 --stagger 0 --no-cpu-baseline --dtype f16|f16r` (9x9, 8192-leaf "
                 "launches, M = 663552 rows); medians over the full-batch launches of k_conv3x3_h2<9,256,256,EPI>; final round-2 build "
                 "(chunk-major fp16 activations, counted residual wait, paired cout mapping)",
       "correction": "FETCH_SIZE doubled (gfx950: MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact; unit KB.  MFMA pipe busy = "
                     "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs), the round-1 convention", "modes": {}}
M, F = 8192 * 81, 256
for D in ("f16", "f16r"):
    sq, fe, wr = load(f"{ROOT}/sq_{D}"), load(f"{ROOT}/fetch_{D}"), load(f"{ROOT}/write_{D}")
    res = {}
    for k in sorted(sq):
        gmax = max(x[2] for x in sq[k]["GRBM_GUI_ACTIVE"])
        med = lambda per, c: statistics.median([x[0] for x in per[k][c] if x[2] == gmax])
        dur = statistics.median([x[1] for x in sq[k]["GRBM_GUI_ACTIVE"] if x[2] == gmax])
        mf, gui, wc, wa, f, w = med(sq, "SQ_VALU_MFMA_BUSY_CYCLES"), med(sq, "GRBM_GUI_ACTIVE"), med(sq, "SQ_WAVE_CYCLES"), med(sq, "SQ_WAIT_ANY"), med(fe, "FETCH_SIZE"), med(wr, "WRITE_SIZE")
        name = {"0": "EPI 0 (conv1: relu -> h16)", "1": "EPI 1 (conv2: + residual, writes residual stream + next activation)"}[k]
        res[name] = {"launch_us_median_under_pmc": round(dur / 1e3, 1), "mfma_pipe_busy": round(mf / (gui / 8 * 1024), 3),
                     "clock_ghz_effective": round(gui / 8 / dur, 3), "tflops_under_pmc": round(2 * 9 * F * F * M / dur / 1e3, 1),
                     "wave_cycles_waiting_share": round(wa / wc, 3), "FETCH_SIZE_KB_raw": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
                     "hbm_bytes_per_launch": round((2 * f + w) * 1024), "hbm_tb_per_s": round((2 * f + w) * 1024 / dur / 1e3, 2)}
    line = json.loads(open(f"{ROOT}/line_{D}.json").read())
    out["modes"][D] = {"kernels": res, "bench_conv_tflops_unprofiled": line["roofline"]["achieved"], "bench_frac_of_2500": line["roofline"]["frac"],
                       "avg_launch_ms_unprofiled": line["roofline"]["avg_launch_ms"]}
json.dump(out, open("profiles/r2_pmc_f16_conv256.json", "w"), indent=1)
for D, m in out["modes"].items():
    print(D, m["bench_conv_tflops_unprofiled"], {k[:5]: (v["launch_us_median_under_pmc"], v["mfma_pipe_busy"], v["hbm_tb_per_s"], v["wave_cycles_waiting_share"]) for k, v in m["kernels"].items()})
