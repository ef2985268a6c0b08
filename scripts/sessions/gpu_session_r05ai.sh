#!/bin/bash
# round 5, session ai: which zero fills / copies a training step still issues (rocprofv3 kernel trace -> rocprof_summary.py fills), training bench on the current tree
cd "$(dirname "$0")/.."
ulimit -c 0
export TMPDIR=/tmp
O=gpurun_out/r05ai; mkdir -p $O
timeout -k 5 300 rocprofv3 --kernel-trace -d $O/tr -o t -- python bench.py --mode train --steps 3 --warmup 2 --no-cpu-baseline > $O/under.json 2> $O/tr.log
python scripts/rocprof_summary.py fills "$(find $O/tr -name '*.db' | head -1)" stem_conv7x7 $O/fills.txt
cat $O/fills.txt
rm -rf $O/tr
timeout 300 python bench.py --mode train --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05ai/bench_train.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
PY
