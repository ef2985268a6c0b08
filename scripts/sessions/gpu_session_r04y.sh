#!/bin/bash
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r04y; mkdir -p $O
CERB_PP_SERIAL_FLOODS=1 timeout 120 python scripts/dev_pp_nuclei_only.py 8192 2>&1 | tail -1
CERB_PP_SERIAL_FLOODS=1 timeout -k 5 200 rocprofv3 --kernel-trace -d $O/ptrace -o p -- python scripts/dev_pp_nuclei_only.py 8192 > $O/pp.log 2>&1
python scripts/rocprof_summary.py timeline "$(find $O/ptrace -name '*.db' | head -1)" nuc_threshold $O/serial_timeline.txt
rm -rf $O/ptrace
grep "ws_flood" $O/serial_timeline.txt
