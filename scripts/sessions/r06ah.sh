#!/bin/bash
O=gpurun_out/r06ah; mkdir -p $O
timeout 900 python -m pytest tests/test_drivers_gpu.py -q -x -k "stored_finer or pipelined" 2>&1 | tail -25 > $O/pytest.txt; cat $O/pytest.txt
