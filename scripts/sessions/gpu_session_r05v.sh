#!/bin/bash
# round 5, session v: packed work items of conv_wino4b (28^2 / 56^2 maps): bitwise A/B tests, training A/B, training bench
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05v; mkdir -p $O
timeout 900 python -m pytest tests/test_net_gpu.py tests/test_train_loss_gpu.py -x -q -m gpu -k "packed or golden or statistics_from_the_conv or backward_pass or crop" 2>&1 | tail -25 > $O/tests.log
cat $O/tests.log
timeout 300 python bench.py --mode train --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05v/bench_train.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for k in d['kernels'][:12]: print(k['kernel'], k['launches'], k['ms_per_step'], k.get('frac'))
PY
CERB_W4B_PACKED=0 timeout 300 python bench.py --mode train --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_train_blocks.json 2> $O/bench_train_blocks.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05v/bench_train_blocks.json').read().strip().splitlines()[-1])
print('blocks:', d['value'], d['ms_per_step'])
for k in d['kernels'][:12]: print(k['kernel'], k['launches'], k['ms_per_step'], k.get('frac'))
PY
