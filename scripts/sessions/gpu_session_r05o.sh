#!/bin/bash
# round 5, session o: nuclei front with root bitmaps (ranks, per-component setup, heap ranges, work lists) -- parity + timeline, A/B against CERB_PP_ONE_PIXEL_THREADS=1 (one pixel per thread) and CERB_PP_PIXEL_SCANS=1 (round 4's whole-map scans)
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05o; mkdir -p $O
timeout 900 python -m pytest tests/test_postproc_gpu.py -q -m gpu -x 2>&1 | tail -8 > $O/pp_tests.log
cat $O/pp_tests.log
for mode in bitmaps seamstrict narrow pixel; do
  unset CERB_PP_PIXEL_SCANS CERB_PP_ONE_PIXEL_THREADS CERB_PP_SEAM_STRICT
  if [ $mode = seamstrict ]; then export CERB_PP_SEAM_STRICT=1; fi
  if [ $mode = pixel ]; then export CERB_PP_PIXEL_SCANS=1; fi
  if [ $mode = narrow ]; then export CERB_PP_ONE_PIXEL_THREADS=1; fi
  timeout 200 python scripts/dev_pp_nuclei_only.py 8192 > $O/pp_$mode.log 2>&1; tail -2 $O/pp_$mode.log
  timeout -k 5 200 rocprofv3 --kernel-trace -d $O/ptrace_$mode -o p -- python scripts/dev_pp_nuclei_only.py 8192 > $O/ptrace_$mode.log 2>&1
  python scripts/rocprof_summary.py timeline "$(find $O/ptrace_$mode -name '*.db' | head -1)" nuc_threshold $O/timeline_$mode.txt
  rm -rf $O/ptrace_$mode
done
unset CERB_PP_PIXEL_SCANS CERB_PP_ONE_PIXEL_THREADS CERB_PP_SEAM_STRICT
head -30 $O/timeline_bitmaps.txt | cut -c1-110
grep -n 'seam\|flatten_roots' $O/timeline_seamstrict.txt | cut -c1-110
CERB_PP_SEAM_STRICT=1 timeout 900 python -m pytest tests/test_postproc_gpu.py -q -m gpu -x 2>&1 | tail -2
