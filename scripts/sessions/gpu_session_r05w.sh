#!/bin/bash
# round 5, session w: the packed-items training A/B test; how conv_wgrad_wino's launch time follows the bytes a chunk pulls (ablations); upadd_bwd / maxpool_bwd after the XCD order
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05w; mkdir -p $O
timeout 600 python -m pytest tests/test_train_loss_gpu.py -x -q -m gpu -k "packed" 2>&1 | tail -15 > $O/tests.log
cat $O/tests.log
timeout 200 python scripts/dev_train_layers.py "upadd_bwd" > $O/upadd.txt 2>&1; tail -6 $O/upadd.txt
timeout 200 python scripts/dev_train_layers.py "maxpool_bwd" > $O/maxpool.txt 2>&1; tail -3 $O/maxpool.txt
bash scripts/dev_wwabl.sh ";-DWW_ABL_HALFX;-DWW_ABL_NOX;-DWW_ABL_NOY;-DWW_ABL_HALFX -DWW_ABL_NOE;-DWW_ABL_HALFX -DWW_ABL_NOM" > $O/wwabl.txt 2>&1
cat $O/wwabl.txt
