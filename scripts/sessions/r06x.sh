#!/bin/bash
O=gpurun_out/r06x; mkdir -p $O
timeout 1500 python -m pytest tests/test_cli_gpu.py -q -x -k "mask" 2>&1 | tail -30 > $O/pytest_mask.txt; cat $O/pytest_mask.txt
