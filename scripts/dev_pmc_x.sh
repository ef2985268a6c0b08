#!/bin/bash
# developer tool: LDS counters per kernel for one batch-32 forward
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_x; mkdir -p gpurun_out/pmc_x
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_]*LDS[A-Z_]*" | sort -u | tr "\n" " "; echo
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD \
  -d gpurun_out/pmc_x -o lds -- python scripts/dev_profile_layers.py 32 > gpurun_out/pmc_x/run.log 2>&1
DB=$(find gpurun_out/pmc_x -name "*.db" | head -1)
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
