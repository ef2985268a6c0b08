#!/bin/bash
# round 5, session aj: per-layer table of the training step's F(4x4) forward / data-gradient convolutions; packed-items test with the new geometries
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05aj; mkdir -p $O
timeout 600 python -m pytest tests/test_net_gpu.py -x -q -m gpu -k "packed" 2>&1 | tail -3 > $O/tests.log; cat $O/tests.log
timeout 200 python scripts/dev_train_layers.py "conv_wino4" > $O/layers.txt 2>&1; cat $O/layers.txt | cut -c1-140
