#!/bin/bash
# round 4, session o: timeline of the nuclei labelling at 8192^2 (which launches are on the critical path); wino4s manual waits after the bias fix
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r04o; mkdir -p $O
timeout 120 python scripts/dev_pp_nuclei_only.py 8192 2>&1 | tail -1 | tee $O/pp_nuclei_plain.txt
timeout -k 5 200 rocprofv3 --kernel-trace -d $O/ptrace -o p -- python scripts/dev_pp_nuclei_only.py 8192 > $O/pp_nuclei.log 2>&1
python scripts/rocprof_summary.py timeline "$(find $O/ptrace -name '*.db' | head -1)" nuc_threshold $O/pp_nuclei_timeline.txt
rm -rf $O/ptrace
cp cerberus_amd/csrc/conv_wino4s.o /tmp/w4s_keep.o; cp cerberus_amd/libcerberus_hip.so /tmp/lib_keep.so
CERB_VARIANTS="-DS4_MANUAL_WAITS;-DS4_MANUAL_WAITS -DS4_RING=12 -DS4_WD=11" bash scripts/dev_w4sabl.sh 2>&1 | tee $O/w4s_manual_waits3.txt
cp /tmp/w4s_keep.o cerberus_amd/csrc/conv_wino4s.o; cp /tmp/lib_keep.so cerberus_amd/libcerberus_hip.so
