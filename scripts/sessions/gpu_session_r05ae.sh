#!/bin/bash
# round 5, session ae: Adam table cached (no stream sync per step): training tests + training bench
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05ae; mkdir -p $O
timeout 900 python -m pytest tests/test_train_loss_gpu.py -x -q -m gpu 2>&1 | tail -5 > $O/tests.log
cat $O/tests.log
timeout 300 python bench.py --mode train --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05ae/bench_train.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'].get('attributed_ms'))
for k in d['kernels']:
    if k['ms_per_step']>0.7 or k['kernel'].startswith('('): print(k['kernel'], k['launches'], k['ms_per_step'], k.get('frac'))
PY
