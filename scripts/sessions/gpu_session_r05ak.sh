#!/bin/bash
# round 5, session ak: weight gradients on a side stream: A/B test, training tests, training bench both ways
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05ak; mkdir -p $O
timeout 900 python -m pytest tests/test_train_loss_gpu.py -x -q -m gpu 2>&1 | tail -5 > $O/tests.log
cat $O/tests.log
timeout 300 python bench.py --mode train --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05ak/bench_train.json').read().strip().splitlines()[-1])
print('side:', d['value'], d['ms_per_step'])
PY
CERB_WGRAD_SIDE=0 timeout 300 python bench.py --mode train --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_train_single.json 2> $O/bench_train_single.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05ak/bench_train_single.json').read().strip().splitlines()[-1])
print('single:', d['value'], d['ms_per_step'])
PY
