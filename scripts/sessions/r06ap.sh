#!/bin/bash
O=gpurun_out/r06ap; mkdir -p $O
python -m pytest tests -q -m gpu -x --durations=25 2>&1 | tail -45 > $O/pytest_gpu.txt; tail -40 $O/pytest_gpu.txt
python __graft_entry__.py --smoke 2>&1 | tail -1
