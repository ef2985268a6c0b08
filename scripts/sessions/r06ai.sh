#!/bin/bash
# 40x scans (0.25 / 0.2528 mpp base) of 65536^2 stored pixels read at 0.5 mpp: steady state of decode processes + device reduction
O=gpurun_out/r06ai; mkdir -p $O
for cfg in "0.25 16" "0.2528 16" "0.25 32" "0.25 8"; do set -- $cfg
CERB_DECODE_PROCS=$2 GIANT_BASE_MPP=$1 timeout 1500 python scripts/dev_r06_giant_slide.py 65536 65536 $O/base$1_procs$2.json > $O/h.log 2>&1; echo "mpp $1 procs $2 rc $?"; grep -E "Inference Time|Mpx/s" $O/h.log
done
