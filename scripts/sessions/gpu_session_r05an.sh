#!/bin/bash
# round 5, session an: side-stream fork before / after the layer's data gradient: training bench
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05an; mkdir -p $O
for V in early late early late; do
if [ $V = late ]; then export CERB_WGRAD_FORK_LATE=1; else unset CERB_WGRAD_FORK_LATE; fi
timeout 300 python bench.py --mode train --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_$V.json 2> $O/bench_$V.err
python - $V <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r05an/bench_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], d['ms_per_step'])
PY
done
