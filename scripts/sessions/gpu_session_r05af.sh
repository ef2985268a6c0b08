#!/bin/bash
# round 5, session af: whole GPU suite on the current tree, then the profiles/ refresh (scripts/profile_r05.sh), default bench + 20000^2 bench + training bench lines
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05af; mkdir -p $O
( time timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) > $O/gpu_suite.log 2>&1
cat $O/gpu_suite.log
timeout 600 python bench.py > $O/bench_wsi_40000.json 2> $O/bench_default.err; tail -c 400 $O/bench_wsi_40000.json
timeout 400 python bench.py --slide 20000 --no-train-leg --no-cpu-baseline > $O/bench_wsi_20000.json 2> $O/bench_20000.err; tail -c 200 $O/bench_wsi_20000.json
timeout 300 python bench.py --mode train --steps 10 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err; tail -c 300 $O/bench_train.json
bash scripts/profile_r05.sh > $O/profile.log 2>&1; tail -20 $O/profile.log
