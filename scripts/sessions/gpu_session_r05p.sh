#!/bin/bash
# round 5, session p: post-processing fuzz against the C oracle (random sizes: both the four-pixel and the one-pixel passes), then the whole GPU suite
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05p; mkdir -p $O
make -C oracle -s 2>/dev/null | tail -1
timeout 900 python tests/tools/dev_fuzz_pp.py 300 77 2>&1 | tail -4 > $O/fuzz_a.log; cat $O/fuzz_a.log
timeout 900 python tests/tools/dev_fuzz_pp.py 300 4242 2>&1 | tail -4 > $O/fuzz_b.log; cat $O/fuzz_b.log
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/gpu_suite.log
cat $O/gpu_suite.log
