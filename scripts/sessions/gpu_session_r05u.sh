#!/bin/bash
# round 5, session e (after the container was re-created): the whole GPU suite, smoke, default bench, training bench on HEAD
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05u; mkdir -p $O
( time timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) > $O/gpu_suite.log 2>&1
cat $O/gpu_suite.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
timeout 300 python bench.py --mode train --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err; tail -c 600 $O/bench_train.json
