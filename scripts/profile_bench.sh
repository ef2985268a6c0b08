#!/bin/bash
# Produces the summaries committed under profiles/ (run on the GPU box through gpurun):
#   1. rocprofv3 --kernel-trace --stats of `python bench.py`           -> gpurun_out/prof/kernel_stats.txt
#   2. two separate PMC passes (FETCH_SIZE, WRITE_SIZE; guide section "HBM")  -> gpurun_out/prof/pmc_hbm.json
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python bench.py --slide 12288 --steps 20 --warmup 5 > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
python scripts/rocprof_summary.py stats "$(find $OUT/stats -name '*.db' | head -1)" $OUT/kernel_stats.txt
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o f -- python bench.py --mode batch --no-cpu-baseline --steps 5 --warmup 1 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o w -- python bench.py --mode batch --no-cpu-baseline --steps 5 --warmup 1 > $OUT/write.log 2>&1
python scripts/rocprof_summary.py pmc "$(find $OUT/fetch -name '*.db' | head -1)" "$(find $OUT/write -name '*.db' | head -1)" $OUT/pmc_hbm.json
rm -rf $OUT/stats $OUT/fetch $OUT/write
tail -1 $OUT/bench_under_rocprof.json | cut -c1-300
head -12 $OUT/kernel_stats.txt
