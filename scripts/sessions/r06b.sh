#!/bin/bash
# round 6: planted-batch guard test + rocpd schema probe (kernels / counters_collection views) for the grid-grouped summaries
O=gpurun_out/r06b; mkdir -p $O
python -m pytest tests/test_net_gpu.py -q -m gpu -k "planted or logit_guard" 2>&1 | tail -30 > $O/pytest_net.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o b -- python $GRAFT_REPO_ROOT/bench.py --mode batch --no-cpu-baseline --steps 2 --warmup 1 > /tmp/p1.log 2>&1
timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_MFMA -d /tmp/p2 -o b -- python $GRAFT_REPO_ROOT/bench.py --mode batch --no-cpu-baseline --steps 1 --warmup 1 > /tmp/p2.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY' > $O/schema.txt 2>&1
import sqlite3, glob
for d in ("/tmp/p1", "/tmp/p2"):
    db = glob.glob(d + "/**/*.db", recursive=True)[0]
    c = sqlite3.connect(db)
    print("==", db)
    for name, typ, sql in c.execute("select name, type, sql from sqlite_master where type in ('view') order by name"):
        if name in ("kernels", "top_kernels", "counters_collection", "pmc_events", "kernel_summary"):
            print(name, typ, sql)
    for v in ("kernels", "counters_collection"):
        try:
            cur = c.execute("select * from %s limit 2" % v)
            print(v, [x[0] for x in cur.description])
            for r in cur:
                print("   ", r)
        except Exception as e:
            print(v, "ERR", e)
PY
tail -3 $O/pytest_net.txt; head -c 6000 $O/schema.txt
