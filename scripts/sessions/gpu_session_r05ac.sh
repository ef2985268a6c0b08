#!/bin/bash
# round 5, session ac: conv_wgrad_wino, one load behind every matrix instruction (1 : 1, 3 : 2, 2 : 1) against nine behind each of six positions; 8 / 10 / 11 per position
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05ac; mkdir -p $O
bash scripts/dev_wwabl.sh ";-DWW_FINE -DWW_FINE_NUM=1 -DWW_FINE_DEN=1;-DWW_FINE -DWW_FINE_NUM=3 -DWW_FINE_DEN=2;-DWW_FINE -DWW_FINE_NUM=2 -DWW_FINE_DEN=1;-DWW_LPP=8;-DWW_LPP=10;-DWW_LPP=11" > $O/wwabl.txt 2>&1
cat $O/wwabl.txt
