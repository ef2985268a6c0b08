#!/bin/bash
# round 4, session p: nuclei labelling passes (4-pixel threshold / erode, run-aggregated counts, boxes only for flooding components)
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r04p; mkdir -p $O
timeout 120 python scripts/dev_pp_nuclei_only.py 8192 2>&1 | tail -1 | tee $O/pp_nuclei_plain.txt
timeout 1200 python -m pytest tests/test_postproc_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pp_tests.txt
timeout -k 5 200 rocprofv3 --kernel-trace -d $O/ptrace -o p -- python scripts/dev_pp_nuclei_only.py 8192 > $O/pp_nuclei.log 2>&1
python scripts/rocprof_summary.py timeline "$(find $O/ptrace -name '*.db' | head -1)" nuc_threshold $O/pp_nuclei_timeline.txt
rm -rf $O/ptrace
