#!/bin/bash
O=gpurun_out/r06q; mkdir -p $O
python tests/tools/dev_fuzz_stream_ranks.py 20 99 2>&1 | grep -E "case|fuzz" | tee $O/fuzz_stream_ranks.txt

