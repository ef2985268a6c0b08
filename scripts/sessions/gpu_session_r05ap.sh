#!/bin/bash
# round 5, session ap: where conv_wino4b hands over to conv_wino4 (CERB_W4B_MAX_PX): 4096 (default) / 12544 (the 112^2 maps of a 448-pixel patch on one-block items) / 50176: training bench
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05ap; mkdir -p $O
for V in 4096 12544 50176 4096 12544; do
CERB_W4B_MAX_PX=$V timeout 300 python bench.py --mode train --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_$V.json 2> $O/bench_$V.err
python - $V <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r05ap/bench_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], d['ms_per_step'], [(k['kernel'],k['launches'],k['ms_per_step']) for k in d['kernels'] if 'conv_wino4' in k['kernel']])
PY
done
