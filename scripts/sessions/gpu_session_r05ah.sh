#!/bin/bash
# round 5, session ah: conv_wgrad_wino A/B: default (loads of chunk i + 1 behind six positions) / with scheduling barriers / gradient loads one more chunk ahead
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05ah; mkdir -p $O
bash scripts/dev_wwabl.sh ";-DWW_SCHEDB;-DWW_DY_AHEAD -DWW_SCHEDB;-DWW_DY_AHEAD" > $O/wwabl.txt 2>&1
cat $O/wwabl.txt
