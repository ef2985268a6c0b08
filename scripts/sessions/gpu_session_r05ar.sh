#!/bin/bash
# round 5, session ar: final check of the tree: whole GPU suite, smoke, default bench line
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05ar; mkdir -p $O
( time timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) > $O/gpu_suite.log 2>&1
cat $O/gpu_suite.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
( time timeout 600 python bench.py > $O/bench_wsi_40000.json 2> $O/bench_default.err ) 2>&1 | grep real; tail -c 300 $O/bench_wsi_40000.json
timeout 300 python bench.py --mode train --steps 10 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err; tail -c 200 $O/bench_train.json
