#!/bin/bash
# round 5, session r: one training step as a dispatch timeline (which zero fills / copies does a step still issue, and whose are they)
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05r; mkdir -p $O
timeout -k 5 400 rocprofv3 --kernel-trace -d $O/ttrace -o t -- python bench.py --mode train --steps 2 --warmup 2 --no-cpu-baseline > $O/train.json 2> $O/train.err
DB=$(find $O/ttrace -name '*.db' | head -1)
python scripts/rocprof_summary.py fills "$DB" "stem_conv7x7_kernel<false>" $O/train_fills.txt
python scripts/rocprof_summary.py timeline "$DB" "stem_conv7x7_kernel<false>" $O/train_timeline.txt
rm -rf $O/ttrace
cat $O/train_fills.txt | cut -c1-150
