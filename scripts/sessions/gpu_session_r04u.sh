#!/bin/bash
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r04u; mkdir -p $O
timeout 900 python -m pytest tests/test_drivers_gpu.py tests/test_cli_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 200 $O/bench.json
