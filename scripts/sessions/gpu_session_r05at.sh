#!/bin/bash
# round 5, session at: tree with the BatchNorm backward sums in the data gradients: training tests, network tests, training bench
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05at; mkdir -p $O
timeout 1200 python -m pytest tests/test_train_loss_gpu.py tests/test_net_gpu.py -x -q -m gpu 2>&1 | tail -4 > $O/tests.log
cat $O/tests.log
for V in fused pass fused; do
if [ $V = pass ]; then export CERB_BN_BWD_PASS1=1; else unset CERB_BN_BWD_PASS1; fi
timeout 300 python bench.py --mode train --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_$V.json 2> $O/bench_$V.err
python - $V <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r05at/bench_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], d['ms_per_step'])
PY
done
