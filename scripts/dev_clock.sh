#!/bin/bash
# developer tool: the shader clock a kernel actually runs at = GRBM_GUI_ACTIVE cycles / dispatch duration (one PMC pass of the batch step)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_clk; mkdir -p gpurun_out/pmc_clk
timeout -k 5 150 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/pmc_clk -o p -- python bench.py --mode batch --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/pmc_clk/run.log 2>&1
DB=$(find gpurun_out/pmc_clk -name "*.db" | head -1)
python - "$DB" "${1:-conv_wino4p|head_group|conv_wino4b}" <<'PY'
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
try:
    rows = list(c.execute("select kernel_name, value, (end - start) from counters_collection where counter_name='GRBM_GUI_ACTIVE'"))
except Exception as e:
    print("schema:", names)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    print("counters_collection columns:", cols)
    raise
agg = {}
for k, v, dur in rows:
    if re.search(sys.argv[2], k) and dur:
        agg.setdefault(k[:60], []).append((v, dur))
for k, l in agg.items():
    cyc = sum(v for v, _ in l) / len(l)
    ns = sum(d for _, d in l) / len(l)
    print("%-60s n=%3d GRBM_GUI_ACTIVE %.4g cycles / %.1f us = %.3f GHz" % (k, len(l), cyc, ns / 1e3, cyc / ns))
PY
rm -rf gpurun_out/pmc_clk
