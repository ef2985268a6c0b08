#!/bin/bash
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r04x; mkdir -p $O
for cfg in "64 2" "64 1"; do set -- $cfg
  timeout -k 5 200 rocprofv3 --kernel-trace -d $O/tr_$1_$2 -o t -- python scripts/dev_two_streams.py $1 $2 > $O/run_$1_$2.log 2>&1
  python scripts/rocprof_summary.py busy "$(find $O/tr_$1_$2 -name '*.db' | head -1)" $O/busy_$1_$2.txt
  rm -rf $O/tr_$1_$2
  tail -2 $O/run_$1_$2.log | head -1; cat $O/busy_$1_$2.txt
done
