#!/bin/bash
# fused attention: parity, fused vs pair lines, then the stamp build (copied over the product library in this scratch copy only)
bash scripts/r3_x3_att.sh || exit 1
bash scripts/r3_att_stamp.sh
