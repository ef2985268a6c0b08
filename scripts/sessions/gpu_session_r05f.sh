#!/bin/bash
# round 5, session f: the new tests (streaming, world-1 RCCL) then the whole GPU suite on the current tree
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05f; mkdir -p $O
timeout 1500 python -m pytest tests/test_drivers_gpu.py tests/test_cli_gpu.py tests/test_train_loss_gpu.py -x -q -m gpu -k "streamed or streams_a_slide or nccl or winograd_domain or reference_golden or slide_20000" 2>&1 | tail -25 > $O/new_tests.log
cat $O/new_tests.log
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/gpu_suite.log
cat $O/gpu_suite.log
