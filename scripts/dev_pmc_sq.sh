#!/bin/bash
# developer tool: SQ counters (wait / issue / MFMA busy) per kernel for one batch-32 forward
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_sq; mkdir -p gpurun_out/pmc_sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS \
  -d gpurun_out/pmc_sq -o sq -- python scripts/dev_profile_layers.py 32 > gpurun_out/pmc_sq/run.log 2>&1
DB=$(find gpurun_out/pmc_sq -name "*.db" | head -1)
python - "$DB" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"))
d = {}
for k, n, cnt, v in rows:
    d.setdefault(k[:60], {})[n] = (cnt, v)
for k, m in d.items():
    if 'conv_wino' in k or 'head' in k or 'conv_igemm' in k:
        print(k)
        for n, (cnt, v) in sorted(m.items()):
            print("   %-28s n=%4d avg=%.4g" % (n, cnt, v))
PY
