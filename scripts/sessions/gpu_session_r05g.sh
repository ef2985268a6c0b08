#!/bin/bash
# round 5, session g: two-colour marker labelling (postproc) -- parity suite + timeline; then the rest of the GPU suite
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05g; mkdir -p $O
timeout 1500 python -m pytest tests/test_postproc_gpu.py -x -q -m gpu 2>&1 | tail -8 > $O/pp_tests.log
cat $O/pp_tests.log
for mode in two three; do
  [ $mode = three ] && export CERB_PP_THREE_LABELLINGS=1
  timeout 120 python scripts/dev_pp_nuclei_only.py 8192 2>&1 | tail -1
  timeout -k 5 200 rocprofv3 --kernel-trace -d $O/ptrace_$mode -o p -- python scripts/dev_pp_nuclei_only.py 8192 > $O/pp_$mode.log 2>&1
  python scripts/rocprof_summary.py timeline "$(find $O/ptrace_$mode -name '*.db' | head -1)" nuc_threshold $O/timeline_$mode.txt
  rm -rf $O/ptrace_$mode
  head -42 $O/timeline_$mode.txt
done
unset CERB_PP_THREE_LABELLINGS
timeout 2400 python -m pytest tests -x -q -m gpu --deselect tests/test_postproc_gpu.py 2>&1 | tail -15 > $O/gpu_suite.log
cat $O/gpu_suite.log
