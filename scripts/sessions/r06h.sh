#!/bin/bash
# full GPU suite on the split / dev-switch tree + smoke
O=gpurun_out/r06h; mkdir -p $O
python -m pytest tests -q -m gpu -x 2>&1 | tail -40 > $O/pytest_gpu.txt
tail -15 $O/pytest_gpu.txt
python __graft_entry__.py --smoke 2>&1 | tail -2
