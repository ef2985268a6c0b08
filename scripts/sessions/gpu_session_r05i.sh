#!/bin/bash
# round 5, session i: two-colour labelling with the root list -- parity + timeline
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05i; mkdir -p $O
timeout 1500 python -m pytest tests/test_postproc_gpu.py -x -q -m gpu 2>&1 | tail -8 > $O/pp_tests.log
cat $O/pp_tests.log
timeout 120 python scripts/dev_pp_nuclei_only.py 8192 2>&1 | tail -1
timeout -k 5 200 rocprofv3 --kernel-trace -d $O/ptrace -o p -- python scripts/dev_pp_nuclei_only.py 8192 > $O/pp.log 2>&1
python scripts/rocprof_summary.py timeline "$(find $O/ptrace -name '*.db' | head -1)" nuc_threshold $O/timeline.txt
rm -rf $O/ptrace
head -40 $O/timeline.txt
