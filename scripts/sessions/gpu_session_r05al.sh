#!/bin/bash
# round 5, session al: side stream priority (default / low / high): training bench
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05al; mkdir -p $O
for P in default low high default; do
CERB_WGRAD_SIDE_PRIO=$P timeout 300 python bench.py --mode train --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_$P.json 2> $O/bench_$P.err
python - $P <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r05al/bench_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], d['ms_per_step'])
PY
done
