#!/bin/bash
# whole GPU suite (no -x: every failure in one pass) after the bench warm-up fix, with durations
O=gpurun_out/r06bb; mkdir -p $O
S=$(date +%s)
timeout 1150 python -m pytest tests -q -m gpu --durations=45 > $O/pytest_gpu.txt 2>&1; echo "pytest rc $? in $(( $(date +%s) - S )) s" | tee -a $O/pytest_gpu.txt
tail -75 $O/pytest_gpu.txt
