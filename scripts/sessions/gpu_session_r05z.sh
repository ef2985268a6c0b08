#!/bin/bash
# round 5, session z: conv_wgrad_wino -- do the matrix phase's LDS operand reads hold back the global loads in flight?
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05z; mkdir -p $O
bash scripts/dev_wwabl.sh ";-DWW_ABL_NOLDSR;-DWW_ABL_NOLDSR -DWW_ABL_NOLOAD;-DWW_ABL_NOLDSR -DWW_ABL_NOE" > $O/wwabl.txt 2>&1
cat $O/wwabl.txt
