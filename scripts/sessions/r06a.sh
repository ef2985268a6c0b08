#!/bin/bash
# round 6, first GPU contact: new precision fixtures + logit guard + bench line with conv_algo legs
O=gpurun_out/r06a; mkdir -p $O
python -m pytest tests/test_net_gpu.py -q -m gpu -k "planted" -s 2>&1 | tail -60 > $O/pytest_net.txt

tail -5 $O/pytest_net.txt
