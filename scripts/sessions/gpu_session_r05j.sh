#!/bin/bash
# round 5, session j: per-layer BatchNorm records of the training step (where do the 26 ms go?)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r05j; mkdir -p $O
timeout 300 python scripts/dev_train_layers.py bn_ > $O/bn_layers.txt 2>&1
sort -k3 -n -r $O/bn_layers.txt | head -40
grep total $O/bn_layers.txt
