#!/bin/bash
# round 5, session am: training kernel statistics on one stream (the rocprof summary under profiles/), as scripts/profile_r05.sh now takes them
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
OUT=gpurun_out/prof_r05b; rm -rf $OUT; mkdir -p $OUT
CERB_WGRAD_SIDE=0 timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $OUT/tstats -o t -- python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $OUT/r05_bench_train_under_rocprof.json 2> $OUT/tstats.log
python scripts/rocprof_summary.py stats "$(find $OUT/tstats -name '*.db' | head -1)" $OUT/r05_bench_train_kernel_stats.txt
rm -rf $OUT/tstats
head -12 $OUT/r05_bench_train_kernel_stats.txt | cut -c1-150
