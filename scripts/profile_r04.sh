#!/bin/bash
# Round-4 summaries for profiles/ (run on the GPU box through gpurun; every rocprofv3 run is wrapped in its own timeout, counters in their own passes):
#   r04_batch32_kernel_stats.txt   rocprofv3 --kernel-trace --stats of `bench.py --mode batch` (BASELINE.json configs[1]; one symbol per launch size)
#   r04_bench_kernel_stats.txt     the same of the default `bench.py` on a 12288^2 slide
#   r04_bench_pmc_hbm.json         FETCH_SIZE / WRITE_SIZE in two separate passes (guide section "HBM")
#   r04_bench_train_kernel_stats.txt   of `bench.py --mode train`
#   r04_postproc_nuclei_8192_kernel_stats.txt   of scripts/dev_pp_nuclei_only.py 8192 (the latest post-proc profile was round 1's)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_r04
rm -rf $OUT; mkdir -p $OUT
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/bstats -o b -- python bench.py --mode batch --no-cpu-baseline --steps 20 --warmup 3 > $OUT/r04_batch32_under_rocprof.json 2> $OUT/bstats.log
python scripts/rocprof_summary.py stats "$(find $OUT/bstats -name '*.db' | head -1)" $OUT/r04_batch32_kernel_stats.txt
timeout -k 5 500 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python bench.py --slide 12288 --steps 20 --warmup 5 > $OUT/r04_bench_under_rocprof.json 2> $OUT/stats.log
python scripts/rocprof_summary.py stats "$(find $OUT/stats -name '*.db' | head -1)" $OUT/r04_bench_kernel_stats.txt
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o f -- python bench.py --mode batch --no-cpu-baseline --steps 5 --warmup 1 > $OUT/fetch.log 2>&1
timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o w -- python bench.py --mode batch --no-cpu-baseline --steps 5 --warmup 1 > $OUT/write.log 2>&1
python scripts/rocprof_summary.py pmc "$(find $OUT/fetch -name '*.db' | head -1)" "$(find $OUT/write -name '*.db' | head -1)" $OUT/r04_bench_pmc_hbm.json
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $OUT/tstats -o t -- python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $OUT/r04_bench_train_under_rocprof.json 2> $OUT/tstats.log
python scripts/rocprof_summary.py stats "$(find $OUT/tstats -name '*.db' | head -1)" $OUT/r04_bench_train_kernel_stats.txt
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $OUT/pstats -o p -- python scripts/dev_pp_nuclei_only.py 8192 > $OUT/pp_nuclei.log 2>&1
python scripts/rocprof_summary.py stats "$(find $OUT/pstats -name '*.db' | head -1)" $OUT/r04_postproc_nuclei_8192_kernel_stats.txt
rm -rf $OUT/stats $OUT/bstats $OUT/fetch $OUT/write $OUT/tstats $OUT/pstats
head -30 $OUT/r04_batch32_kernel_stats.txt
tail -1 $OUT/r04_bench_under_rocprof.json | cut -c1-300
head -40 $OUT/r04_postproc_nuclei_8192_kernel_stats.txt | cut -c1-140
cat $OUT/pp_nuclei.log | tail -2
