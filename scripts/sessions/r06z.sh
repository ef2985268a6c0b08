#!/bin/bash
# the command-line and driver suites after the between-slides release, the budget change and the masked banded nuclei
O=gpurun_out/r06z; mkdir -p $O
timeout 2400 python -m pytest tests/test_cli_gpu.py tests/test_drivers_gpu.py tests/test_postproc_gpu.py -q -x 2>&1 | tail -8 > $O/pytest.txt; cat $O/pytest.txt
