#!/bin/bash
# round 5, session ag: conv_wgrad_wino with the gradient loads one more chunk ahead + coalesced reduce; stem_wgrad's parallel reduce: training tests, per-layer times
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05ag; mkdir -p $O
timeout 900 python -m pytest tests/test_train_loss_gpu.py -x -q -m gpu 2>&1 | tail -5 > $O/tests.log
cat $O/tests.log
timeout 200 python scripts/dev_train_layers.py wgrad_wino4 > $O/ww.txt 2>&1; grep -E "dec\.3\.1|dec\.2\.1|dec\.0\.0|layer1.0.conv1|layer3.1.conv1|layer4.1.conv1|total" $O/ww.txt
timeout 200 python scripts/dev_train_layers.py stem_wgrad > $O/stem.txt 2>&1; tail -2 $O/stem.txt
timeout 300 python bench.py --mode train --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05ag/bench_train.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for k in d['kernels'][:3]: print(k['kernel'], k['launches'], k['ms_per_step'], k.get('frac'))
PY
