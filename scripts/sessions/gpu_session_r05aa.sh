#!/bin/bash
# round 5, session aa: conv_wgrad_wino with its loads spread between the matrix instructions: A/B tests against the direct kernel, per-layer times, bulk-fetch build beside it
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05aa; mkdir -p $O
timeout 900 python -m pytest tests/test_train_loss_gpu.py -x -q -m gpu -k "winograd_domain or backward_pass or whole_train_step" 2>&1 | tail -8 > $O/tests.log
cat $O/tests.log
bash scripts/dev_wwabl.sh ";-DWW_BULK_FETCH;-DWW_ABL_NOE;-DWW_ABL_NOLOAD" > $O/wwabl.txt 2>&1
cat $O/wwabl.txt
