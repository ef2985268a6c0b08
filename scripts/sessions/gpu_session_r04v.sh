#!/bin/bash
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r04v; mkdir -p $O
timeout 900 python -m pytest tests/test_train_loss_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
timeout 600 python bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline > $O/train.json 2> $O/train.err; tail -c 300 $O/train.err
