#!/bin/bash
# developer tool: arbitrary PMC counters per kernel for the configs[1] batch step
#   scripts/dev_pmc_any.sh "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY ..." [kernel-name filter regex]     (several passes: separate the sets with ';')
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
FILTER="${2:-conv_wino}"
IFS=';'
i=0
for SET in $1; do
  unset IFS
  i=$((i+1))
  rm -rf gpurun_out/pmc_any; mkdir -p gpurun_out/pmc_any
  timeout -k 5 150 rocprofv3 --pmc $SET -d gpurun_out/pmc_any -o p -- python bench.py --mode batch --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/pmc_any/run.log 2>&1
  DB=$(find gpurun_out/pmc_any -name "*.db" | head -1)
  [ -z "$DB" ] && { echo "pass $i: no database"; tail -5 gpurun_out/pmc_any/run.log; IFS=';'; continue; }
  python - "$DB" "$FILTER" <<'PY'
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"))
d = {}
for k, n, cnt, v in rows:
    d.setdefault(k[:70], {})[n] = (cnt, v)
for k, m in d.items():
    if re.search(sys.argv[2], k):
        print(k)
        for n, (cnt, v) in sorted(m.items()):
            print("   %-36s n=%4d avg=%.5g" % (n, cnt, v))
PY
  rm -rf gpurun_out/pmc_any
  IFS=';'
done
