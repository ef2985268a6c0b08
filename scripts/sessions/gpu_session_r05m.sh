#!/bin/bash
# round 5, session m: the whole GPU suite on the current tree
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05m; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -40 > $O/gpu_suite.log
cat $O/gpu_suite.log
