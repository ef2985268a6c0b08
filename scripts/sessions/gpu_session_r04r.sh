#!/bin/bash
# round 4, session r: default bench after the fast .dat writer; host-logic dat tests on the box's numpy
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r04r; mkdir -p $O
timeout 300 python -m pytest tests/test_host_logic.py -q -k dat 2>&1 | tail -2 | tee $O/dat_tests.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
