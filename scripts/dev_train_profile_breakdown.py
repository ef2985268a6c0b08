"""Developer: kernel breakdown of the LAST gradient step in a rocprofv3 --kernel-trace database of tests/tools/dev_train_step_time.py."""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
stems = [i for i, r in enumerate(rows) if 'stem_conv7x7' in r[0]]
seg = rows[stems[-2]:stems[-1]]  # last train_grads call (the forward-only call follows it)
print("wall %.1f ms, busy %.1f ms, %d dispatches" % ((seg[-1][2] - seg[0][1]) / 1e6, sum(r[2] - r[1] for r in seg) / 1e6, len(seg)))
agg = defaultdict(lambda: [0, 0.0])
for n, s, e in seg:
    agg[n[:64]][0] += 1; agg[n[:64]][1] += (e - s) / 1e3
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print("%-66s calls %4d total %9.1f us avg %8.1f" % (k, v[0], v[1], v[1] / v[0]))
