#!/bin/bash
# round 5, session ad: head_bwd2 with its fetch spread between the matrix instructions: head A/B tests, times against the bulk fetch
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05ad; mkdir -p $O
timeout 900 python -m pytest tests/test_train_loss_gpu.py -x -q -m gpu -k "fused_output_heads or backward_pass or whole_train_step or subtype" 2>&1 | tail -8 > $O/tests.log
cat $O/tests.log
bash scripts/dev_htabl.sh ";-DHB2_BULK_FETCH" > $O/htabl.txt 2>&1
cat $O/htabl.txt
