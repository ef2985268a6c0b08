#!/bin/bash
# fuzz of the post-processing against the C oracle (compact seam columns), then the full GPU suite
O=gpurun_out/r06l; mkdir -p $O
python tests/tools/dev_fuzz_pp.py 300 4242 2>&1 | tail -4 > $O/fuzz_pp.txt; cat $O/fuzz_pp.txt
python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
python __graft_entry__.py --smoke 2>&1 | tail -1
