#!/bin/bash
# developer tool: PMC counters per kernel for the configs[4] training step
#   scripts/dev_pmc_train.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES ...;<second pass>" [kernel-name filter regex]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
FILTER="${2:-wgrad_wino}"
IFS=';'
i=0
for SET in $1; do
  unset IFS
  i=$((i+1))
  rm -rf gpurun_out/pmc_train; mkdir -p gpurun_out/pmc_train
  timeout -k 5 300 rocprofv3 --pmc $SET -d gpurun_out/pmc_train -o p -- python bench.py --mode train --no-cpu-baseline --steps 1 --warmup 1 > gpurun_out/pmc_train/run.log 2>&1
  DB=$(find gpurun_out/pmc_train -name "*.db" | head -1)
  [ -z "$DB" ] && { echo "pass $i: no database"; tail -5 gpurun_out/pmc_train/run.log; IFS=';'; continue; }
  python - "$DB" "$FILTER" <<'PY'
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection group by kernel_name, counter_name"))
d = {}
for k, n, cnt, v, s in rows:
    d.setdefault(k[:90], {})[n] = (cnt, v, s)
for k, m in d.items():
    if re.search(sys.argv[2], k):
        print(k)
        for n, (cnt, v, s) in sorted(m.items()):
            print("   %-36s n=%4d avg=%.5g sum=%.5g" % (n, cnt, v, s))
PY
  rm -rf gpurun_out/pmc_train
  IFS=';'
done
