#!/bin/bash
O=gpurun_out/r06d; mkdir -p $O
python -m pytest tests/test_cli_gpu.py -q -m gpu -x -k "per_rank_arrays or eight_ranks or two_ranks_equals or nccl_world_one" 2>&1 | tail -40 > $O/pytest_cli.txt
tail -25 $O/pytest_cli.txt
