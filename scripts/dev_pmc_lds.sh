#!/bin/bash
# developer tool: LDS counters per kernel for one batch-32 forward
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_lds; mkdir -p gpurun_out/pmc_lds
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_]*LDS[A-Z_]*" | sort -u | tr "\n" " "; echo
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES \
  -d gpurun_out/pmc_lds -o lds -- python scripts/dev_profile_layers.py 32 > gpurun_out/pmc_lds/run.log 2>&1
DB=$(find gpurun_out/pmc_lds -name "*.db" | head -1)
python - "$DB" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"))
d = {}
for k, n, cnt, v in rows:
    d.setdefault(k[:60], {})[n] = (cnt, v)
for k, m in d.items():
    if 'conv_wino' in k or 'head' in k or 'stem' in k:
        print(k)
        for n, (cnt, v) in sorted(m.items()):
            print("   %-28s n=%4d avg=%.4g" % (n, cnt, v))
PY
