#!/bin/bash
O=gpurun_out/r06c; mkdir -p $O
python -m pytest tests/test_drivers_gpu.py -q -m gpu -x -k "per_rank_instance" 2>&1 | tail -40 > $O/pytest_parts.txt
tail -25 $O/pytest_parts.txt
