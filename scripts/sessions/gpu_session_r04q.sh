#!/bin/bash
# round 4, session q: full GPU suite, smoke, default bench with the current tree
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r04q; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x --durations=12 2>&1 | tail -22 | tee $O/gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
