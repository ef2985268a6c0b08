#!/bin/bash
# how busy is the device during the slide job's inference? (kernel trace of a 16384^2 slide, last second of the trace = steady state + tail)
O=gpurun_out/r06p; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
CERB_AUTO_PRECISION=0 timeout -k 5 400 rocprofv3 --kernel-trace -d /tmp/bz -o b -- python $GRAFT_REPO_ROOT/bench.py --slide 16384 --steps 10 --warmup 2 --no-train-leg --no-ingest-leg --no-cpu-baseline --no-dat --no-ref-tiling > /tmp/bz.json 2> /tmp/bz.err
cd $GRAFT_REPO_ROOT
python - <<'PY' > $O/busy.txt
import sqlite3, glob, json
db = glob.glob('/tmp/bz/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
rows = sorted(c.execute("select start, end, queue_id, name from kernels"))
# the K timed stripes: find the longest window without postproc kernels... simply analyse windows of 1 s across the trace
t0, t1 = rows[0][0], max(r[1] for r in rows)
print("trace %.2f s, %d dispatches" % ((t1 - t0) / 1e9, len(rows)))
import bisect
starts = [r[0] for r in rows]
w = 0.5e9
t = t0
while t + w <= t1:
    i, j = bisect.bisect_left(starts, t), bisect.bisect_left(starts, t + w)
    seg = rows[i:j]
    if seg:
        ev = sorted([(max(a, t), 1) for a, b, q, n in seg] + [(min(b, t + w), -1) for a, b, q, n in seg])
        depth, last, idle = 0, t, 0
        for tt, d in ev:
            if depth == 0: idle += tt - last
            depth += d; last = tt
        if depth == 0: idle += t + w - last
        conv = sum(1 for r in seg if 'conv_wino4p_kernel<1>' in r[3])
        print("window at %6.2f s: idle %5.2f %%  dispatches %5d  conv_wino4p<1> launches %3d" % ((t - t0) / 1e9, 100.0 * idle / w, len(seg), conv))
    t += w
PY
cat $O/busy.txt | tail -40
python - <<'PY'
import json
l=json.loads(open('/tmp/bz.json').read().strip().splitlines()[-1]); print(l['value'], l['config']['inference_Mpx_s'])
PY
