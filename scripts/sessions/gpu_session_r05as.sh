#!/bin/bash
# round 5, session as: BatchNorm backward sums from the data gradient's output stage: A/B test, training tests, training bench both ways
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05as; mkdir -p $O
timeout 900 python -m pytest tests/test_train_loss_gpu.py -x -q -m gpu 2>&1 | tail -12 > $O/tests.log
cat $O/tests.log
for V in fused pass fused pass; do
if [ $V = pass ]; then export CERB_BN_BWD_PASS1=1; else unset CERB_BN_BWD_PASS1; fi
timeout 300 python bench.py --mode train --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_$V.json 2> $O/bench_$V.err
python - $V <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r05as/bench_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], d['ms_per_step'], [(k['kernel'],k['launches'],k['ms_per_step']) for k in d['kernels'] if k['kernel'] in ('bn_bwd','dgrad:conv_wino4<f4x4,16x16x2>','dgrad:conv_wino4b<f4x4,16t>')])
PY
done
