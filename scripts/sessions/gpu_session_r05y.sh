#!/bin/bash
# round 5, session y: upadd_bwd on 2-D workgroup tiles, stem_wgrad with a one-segment prefetch, loss finalize with loads in flight: training tests + per-family times
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05y; mkdir -p $O
timeout 900 python -m pytest tests/test_train_loss_gpu.py -x -q -m gpu 2>&1 | tail -15 > $O/tests.log
cat $O/tests.log
for f in upadd_bwd stem_wgrad head_loss; do timeout 200 python scripts/dev_train_layers.py "$f" > $O/$f.txt 2>&1; tail -7 $O/$f.txt; done
timeout 300 python bench.py --mode train --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05y/bench_train.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for k in d['kernels'][:40]: print(k['kernel'], k['launches'], k['ms_per_step'], k.get('frac'))
PY
