#!/bin/bash
# Round-3 summaries for profiles/ (run on the GPU box through gpurun; every rocprofv3 run is wrapped in its own timeout):
#   batch32_kernel_stats.txt   rocprofv3 --kernel-trace --stats of `bench.py --mode batch` (BASELINE.json configs[1])
#   kernel_stats.txt + bench_under_rocprof.json   the same of the default `bench.py` (whole slide job, --slide 12288 under the profiler)
#   pmc_hbm.json               FETCH_SIZE / WRITE_SIZE in two separate passes (guide section "HBM")
#   sq_counters.txt            SQ / TCP counters of the planar convolution, four passes
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_r03
rm -rf $OUT; mkdir -p $OUT
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/bstats -o b -- python bench.py --mode batch --no-cpu-baseline --steps 20 --warmup 3 > $OUT/batch32_under_rocprof.json 2> $OUT/bstats.log
python scripts/rocprof_summary.py stats "$(find $OUT/bstats -name '*.db' | head -1)" $OUT/batch32_kernel_stats.txt
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python bench.py --slide 12288 --steps 20 --warmup 5 > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
python scripts/rocprof_summary.py stats "$(find $OUT/stats -name '*.db' | head -1)" $OUT/kernel_stats.txt
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o f -- python bench.py --mode batch --no-cpu-baseline --steps 5 --warmup 1 > $OUT/fetch.log 2>&1
timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o w -- python bench.py --mode batch --no-cpu-baseline --steps 5 --warmup 1 > $OUT/write.log 2>&1
python scripts/rocprof_summary.py pmc "$(find $OUT/fetch -name '*.db' | head -1)" "$(find $OUT/write -name '*.db' | head -1)" $OUT/pmc_hbm.json
rm -rf $OUT/stats $OUT/bstats $OUT/fetch $OUT/write
scripts/dev_pmc_any.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY;SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS;TCP_PENDING_STALL_CYCLES TCP_GATE_EN1 TCP_TCC_READ_REQ_sum;TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "conv_wino4p|head_group|upsample2_add_planar" > $OUT/sq_counters.txt 2>&1
head -8 $OUT/batch32_kernel_stats.txt
tail -1 $OUT/bench_under_rocprof.json | cut -c1-300
cat $OUT/sq_counters.txt | head -60
