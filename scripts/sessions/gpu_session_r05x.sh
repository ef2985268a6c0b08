#!/bin/bash
# round 5, session x: max-pool backward by recorded positions (A/B test, timing), the packed-items training test, Dice leg of the bench
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05x; mkdir -p $O
timeout 900 python -m pytest tests/test_train_loss_gpu.py -x -q -m gpu -k "packed or maxpool or backward_pass or whole_train_step" 2>&1 | tail -15 > $O/tests.log
cat $O/tests.log
timeout 200 python scripts/dev_train_layers.py "maxpool" > $O/maxpool.txt 2>&1; tail -4 $O/maxpool.txt
timeout 600 python bench.py --slide 12288 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_small.json 2> $O/bench_small.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05x/bench_small.json').read().strip().splitlines()[-1])
print(d['value'], d.get('dice_vs_reference'), d['train_step'].get('ms_per_step'))
PY
