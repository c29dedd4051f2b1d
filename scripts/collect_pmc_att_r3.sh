#!/bin/bash
# ON THE GPU BOX: PMC passes for the fused attention kernel (raw data stays in /tmp, the summary travels back)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RAW=/tmp/prof_att_raw; rm -rf $RAW; mkdir -p $RAW gpurun_out
B="python3 bench.py --network transgo --dtype f32x3 --steps 1 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $RAW/sq -o sq -- $B > $RAW/sq.log 2>&1 || { echo "sq failed"; tail -5 $RAW/sq.log; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $RAW/fetch -o f -- $B > $RAW/fetch.log 2>&1 || { echo "fetch failed"; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $RAW/write -o w -- $B > $RAW/write.log 2>&1 || { echo "write failed"; exit 1; }
python3 scripts/pmc_att_json.py $RAW gpurun_out/r3_pmc_attention_x3.json
