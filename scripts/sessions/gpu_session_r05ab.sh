#!/bin/bash
# round 5, session ab: conv_wgrad_wino, loads per matrix-phase position: 6 (all nine positions) / 7 / 9 / 13 (front-loaded: the last positions cover the tail's latency)
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05ab; mkdir -p $O
bash scripts/dev_wwabl.sh "-DWW_LPP=6;-DWW_LPP=7;-DWW_LPP=9;-DWW_LPP=13;-DWW_LPP=26" > $O/wwabl.txt 2>&1
cat $O/wwabl.txt
