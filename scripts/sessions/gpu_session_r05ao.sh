#!/bin/bash
# round 5, session ao: visualisation payload through pinned host tensors with one wait: payload test, training bench
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05ao; mkdir -p $O
timeout 900 python -m pytest tests/test_train_loss_gpu.py tests/test_cli_gpu.py -x -q -m gpu -k "raw_payload or whole_train_step or train" 2>&1 | tail -5 > $O/tests.log
cat $O/tests.log
for i in 1 2; do
timeout 300 python bench.py --mode train --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_$i.json 2> $O/bench_$i.err
python - $i <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r05ao/bench_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], [ (k['kernel'],k['ms_per_step']) for k in d['kernels'] if k['kernel'].startswith('(')])
PY
done
