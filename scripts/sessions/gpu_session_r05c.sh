#!/bin/bash
# round 5, session c: out-of-core sub-band streaming + capacity plan (stream_bands.py); then the whole GPU suite on the current tree
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05c; mkdir -p $O
timeout 900 python -m pytest tests/test_drivers_gpu.py tests/test_cli_gpu.py -x -q -m gpu -k "streamed or streams_a_slide or two_handles" 2>&1 | tail -25 > $O/stream_tests.log
cat $O/stream_tests.log
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/gpu_suite.log
cat $O/gpu_suite.log
