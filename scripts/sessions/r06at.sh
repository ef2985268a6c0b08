#!/bin/bash
# kernel statistics of the 40x ingest leg (resample_box_kernel on the copy stream under the inference)
export TMPDIR=/tmp
O=gpurun_out/r06at; mkdir -p $O
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $O/stats -o ing -- python bench.py --mode ingest --slide 12288 --ingest-base-mpp 0.25 > $O/ingest40x_under_rocprof.json 2> $O/stats.log; echo "rc $?"
python scripts/rocprof_summary.py stats "$(find $O/stats -name '*.db' | head -1)" $O/r06_ingest_40x_kernel_stats.txt
grep -E "resample|kernel  " $O/r06_ingest_40x_kernel_stats.txt | cut -c1-200; head -8 $O/r06_ingest_40x_kernel_stats.txt | cut -c1-200
rm -rf $O/stats
