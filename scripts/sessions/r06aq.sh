#!/bin/bash
O=gpurun_out/r06aq; mkdir -p $O
python -m pytest tests/test_cli_gpu.py -q -x --durations=8 -k "nccl_branch or eight_ranks or slide_20000" 2>&1 | tail -16
