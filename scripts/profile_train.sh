#!/bin/bash
# rocprofv3 --kernel-trace --stats of the training bench (python bench.py --mode train) -> gpurun_out/prof_train/kernel_stats.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_train
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_train_under_rocprof.json 2> $OUT/stats.log
python scripts/rocprof_summary.py stats "$(find $OUT/stats -name '*.db' | head -1)" $OUT/kernel_stats.txt
rm -rf $OUT/stats
head -14 $OUT/kernel_stats.txt
