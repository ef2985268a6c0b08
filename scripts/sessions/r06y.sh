#!/bin/bash
O=gpurun_out/r06y; mkdir -p $O
timeout 1500 python -m pytest tests/test_cli_gpu.py -q -x -k "second_slide or mask" 2>&1 | tail -30 > $O/pytest.txt; cat $O/pytest.txt
