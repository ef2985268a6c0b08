#!/bin/bash
# Round-6 summaries for profiles/ (run on the GPU box through gpurun; every rocprofv3 run in its own timeout, counters in their own passes, never with a trace domain):
#   r06_batch32_kernel_stats.txt        rocprofv3 --kernel-trace --stats of `bench.py --mode batch` (BASELINE.json configs[1])
#   r06_bench_kernel_stats.txt          the same of the default `bench.py` on a 12288^2 slide
#   r06_bench_pmc_hbm.json              FETCH_SIZE / WRITE_SIZE of the batch step, two separate passes (guide section "HBM")
#   r06_bench_train_kernel_stats.txt    of `bench.py --mode train` (+ the line it printed: r06_bench_train_under_rocprof.json)
#   r06_train_pmc_hbm.json              FETCH_SIZE / WRITE_SIZE per kernel and STEP of the training leg (bench.py reads it as `traffic`)
#   r06_sq_counters.txt                 SQ_INSTS_MFMA, SQ_VALU_MFMA_BUSY_CYCLES, SQ_WAIT_INST_ANY, TCP / TCC counters for the inference and training kernels
#   r06_postproc_nuclei_8192_kernel_stats.txt, ..._pmc_hbm.json (FETCH_SIZE / WRITE_SIZE per kernel), ..._timeline.txt (dispatches of one call)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
# every profiled command runs WITHOUT the one-tile calibration forward (NetDesc.prepare): each launch of a symbol in a `--mode batch` run is then a batch-32
# launch, and the averages below reproduce bench.py's roofline.frac as written; the summaries also split rows by launch grid and carry median / min / max
export CERB_AUTO_PRECISION=0
OUT=gpurun_out/prof_r06
rm -rf $OUT; mkdir -p $OUT
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/bstats -o b -- python bench.py --mode batch --no-cpu-baseline --steps 20 --warmup 3 > $OUT/r06_batch32_under_rocprof.json 2> $OUT/bstats.log
python scripts/rocprof_summary.py stats "$(find $OUT/bstats -name '*.db' | head -1)" $OUT/r06_batch32_kernel_stats.txt
timeout -k 5 500 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python bench.py --slide 12288 --steps 20 --warmup 5 --no-ingest-leg > $OUT/r06_bench_under_rocprof.json 2> $OUT/stats.log
python scripts/rocprof_summary.py stats "$(find $OUT/stats -name '*.db' | head -1)" $OUT/r06_bench_kernel_stats.txt
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o f -- python bench.py --mode batch --no-cpu-baseline --steps 5 --warmup 1 > $OUT/fetch.log 2>&1
timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o w -- python bench.py --mode batch --no-cpu-baseline --steps 5 --warmup 1 > $OUT/write.log 2>&1
python scripts/rocprof_summary.py pmc "$(find $OUT/fetch -name '*.db' | head -1)" "$(find $OUT/write -name '*.db' | head -1)" $OUT/r06_bench_pmc_hbm.json
# (CERB_WGRAD_SIDE=0: every launch of the step on ONE stream, so that a kernel's duration is its own -- with the weight gradients on their side stream, the default,
#  two kernels share the device and both read longer; the timed figure of the default is in r06_bench_train.json)
CERB_DEV_LIB=1 CERB_WGRAD_SIDE=0 timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $OUT/tstats -o t -- python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $OUT/r06_bench_train_under_rocprof.json 2> $OUT/tstats.log
python scripts/rocprof_summary.py stats "$(find $OUT/tstats -name '*.db' | head -1)" $OUT/r06_bench_train_kernel_stats.txt
# training leg, HBM counters: 1 warm-up + 2 timed + 1 profiled step = 4 steps per run
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/tfetch -o f -- python bench.py --mode train --no-cpu-baseline --steps 2 --warmup 1 > $OUT/tfetch.log 2>&1
timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/twrite -o w -- python bench.py --mode train --no-cpu-baseline --steps 2 --warmup 1 > $OUT/twrite.log 2>&1
python scripts/rocprof_summary.py pmc_step "$(find $OUT/tfetch -name '*.db' | head -1)" "$(find $OUT/twrite -name '*.db' | head -1)" 4 $OUT/r06_train_pmc_hbm.json
# SQ / cache counters: inference kernels (batch step), then training kernels
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA"
SQ2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_MISC"
SQ3="TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
echo "# rocprofv3 --pmc, averages per launch; pass 1: $SQ1 ; pass 2: $SQ2 ; pass 3: $SQ3 (SQ_* in quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES)" > $OUT/r06_sq_counters.txt
i=0
for SET in "$SQ1" "$SQ2" "$SQ3"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $SET -d $OUT/sqb$i -o p -- python bench.py --mode batch --no-cpu-baseline --steps 3 --warmup 1 > $OUT/sqb$i.log 2>&1
  DB=$(find $OUT/sqb$i -name '*.db' | head -1)
  [ -n "$DB" ] && python scripts/rocprof_summary.py sq "$DB" "conv_wino4|head_group|upsample2_add_planar|stem_conv" $OUT/r06_sq_counters.txt append || echo "pass $i (batch): no database: $(tail -2 $OUT/sqb$i.log)" >> $OUT/r06_sq_counters.txt
  timeout -k 5 300 rocprofv3 --pmc $SET -d $OUT/sqt$i -o p -- python bench.py --mode train --no-cpu-baseline --steps 1 --warmup 1 > $OUT/sqt$i.log 2>&1
  DB=$(find $OUT/sqt$i -name '*.db' | head -1)
  [ -n "$DB" ] && python scripts/rocprof_summary.py sq "$DB" "wgrad|bn_bwd|bn_apply|head_fwd|head_bwd|upadd_bwd" $OUT/r06_sq_counters.txt append || echo "pass $i (train): no database: $(tail -2 $OUT/sqt$i.log)" >> $OUT/r06_sq_counters.txt
done
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $OUT/pstats -o p -- python scripts/dev_pp_nuclei_only.py 8192 > $OUT/pp_nuclei.log 2>&1
python scripts/rocprof_summary.py stats "$(find $OUT/pstats -name '*.db' | head -1)" $OUT/r06_postproc_nuclei_8192_kernel_stats.txt
# nuclei labelling, HBM counters per kernel (4 calls per run) + the dispatches of one call in order
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/pfetch -o f -- python scripts/dev_pp_nuclei_only.py 8192 > $OUT/pfetch.log 2>&1
timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/pwrite -o w -- python scripts/dev_pp_nuclei_only.py 8192 > $OUT/pwrite.log 2>&1
python scripts/rocprof_summary.py pmc "$(find $OUT/pfetch -name '*.db' | head -1)" "$(find $OUT/pwrite -name '*.db' | head -1)" $OUT/r06_postproc_nuclei_8192_pmc_hbm.json
# bytes per pixel and pass next to the 12 B/px a call needs (dev_pp_nuclei_only.py makes 4 calls at 8192^2)
python scripts/rocprof_summary.py pmc_table $OUT/r06_postproc_nuclei_8192_pmc_hbm.json 4 67108864 12 $OUT/r06_postproc_nuclei_8192_bytes_per_pass.txt
timeout -k 5 200 rocprofv3 --kernel-trace -d $OUT/ptrace -o p -- python scripts/dev_pp_nuclei_only.py 8192 > $OUT/ptrace.log 2>&1
python scripts/rocprof_summary.py timeline "$(find $OUT/ptrace -name '*.db' | head -1)" nuc_threshold $OUT/r06_postproc_nuclei_8192_timeline.txt
rm -rf $OUT/pfetch $OUT/pwrite $OUT/ptrace
rm -rf $OUT/stats $OUT/bstats $OUT/fetch $OUT/write $OUT/tstats $OUT/pstats $OUT/tfetch $OUT/twrite $OUT/sqb1 $OUT/sqb2 $OUT/sqb3 $OUT/sqt1 $OUT/sqt2 $OUT/sqt3
head -24 $OUT/r06_batch32_kernel_stats.txt
head -30 $OUT/r06_bench_train_kernel_stats.txt | cut -c1-150
tail -1 $OUT/r06_bench_under_rocprof.json | cut -c1-300
wc -l $OUT/r06_sq_counters.txt; head -30 $OUT/r06_sq_counters.txt
