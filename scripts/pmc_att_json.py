#!/usr/bin/env python3
"""Summarise the PMC passes of scripts/collect_profiles.sh att into profiles/r3_pmc_attention_x3.json: MFMA pipe busy, waiting share and
HBM traffic of k_attention_x3<9,128,PRO> (full-batch launches of `bench.py --network transgo --dtype f32x3`)."""
import collections, csv, glob, json, re, statistics, sys
ROOT = sys.argv[1] if len(sys.argv) > 1 else "/tmp/prof_att_raw"
OUT = sys.argv[2] if len(sys.argv) > 2 else "profiles/r3_pmc_attention_x3.json"


def load(d):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_attention_x3ILi9ELi128ELb(\d)E", r["Kernel_Name"]) or re.search(r"k_attention_x3<9, 128, (true|false)>", r["Kernel_Name"])
            if m:
                key = {"0": "PRO=false", "1": "PRO=true", "false": "PRO=false", "true": "PRO=true"}[m.group(1)]
                per[key][r["Counter_Name"]].append((float(r["Counter_Value"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), int(r["Grid_Size"])))
    return per


sq, fe, wr = load(f"{ROOT}/sq"), load(f"{ROOT}/fetch"), load(f"{ROOT}/write")
out = {"source": "rocprofv3 --kernel-trace --pmc ... (separate passes: SQ counters, FETCH_SIZE, WRITE_SIZE; scripts/collect_profiles.sh att) on "
                 "`python3 bench.py --no-launcher --no-cpu-baseline --network transgo --dtype f32x3 --steps 1 --warmup 1` (9x9, 400 sims, MainNetwork 128, 4096 "
                 "boards); medians over the launches of ~16 k boards (duration within 15 % of the largest) of k_attention_x3<9,128,PRO>",
       "correction": "FETCH_SIZE doubled (gfx950: MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact; unit KB.  MFMA pipe busy = "
                     "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs)",
       "algorithmic_bytes_per_board": {"x_read_once": 81 * 128 * 4, "y_write": 81 * 128 * 4, "split_input_write (trunk layers followed by a residual block)": 81 * 128 * 4},
       "kernels": {}}
for k in sorted(sq):
    dmax = max(x[1] for x in sq[k]["GRBM_GUI_ACTIVE"])
    sel = lambda per, c: [x for x in per[k][c] if x[1] >= 0.85 * max(y[1] for y in per[k][c])]
    med = lambda per, c: statistics.median(x[0] for x in sel(per, c)) if per[k].get(c) else None
    ga, mf, wc, wa = med(sq, "GRBM_GUI_ACTIVE"), med(sq, "SQ_VALU_MFMA_BUSY_CYCLES"), med(sq, "SQ_WAVE_CYCLES"), med(sq, "SQ_WAIT_ANY")
    f, w = med(fe, "FETCH_SIZE"), med(wr, "WRITE_SIZE")
    out["kernels"][k] = {"launches_summarised": len(sel(sq, "GRBM_GUI_ACTIVE")),
                         "duration_us_median_under_pmc": round(statistics.median(x[1] for x in sel(sq, "GRBM_GUI_ACTIVE")) / 1e3, 1),
                         "mfma_pipe_busy": round(mf / (ga / 8 * 1024), 4) if mf and ga else None,
                         "wave_cycles_waiting_share": round(wa / wc, 4) if wa and wc else None,
                         "FETCH_SIZE_KB_raw_median": f, "hbm_read_GB": round(2 * f * 1024 / 1e9, 3) if f else None,
                         "WRITE_SIZE_KB_median": w, "hbm_write_GB": round(w * 1024 / 1e9, 3) if w else None}
json.dump(out, open(OUT, "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
