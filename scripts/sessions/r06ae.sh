#!/bin/bash
# a 40x scan (0.25 mpp base, pyramid levels x1 and x16 only) read at 0.5 mpp: the x2 reduction is the reader's job
O=gpurun_out/r06ae; mkdir -p $O
GIANT_BASE_MPP=0.25 timeout 1500 python scripts/dev_r06_giant_slide.py 32768 32768 $O/base025_32768.json > $O/h.log 2>&1; echo "H rc $?"; tail -c 1800 $O/h.log
python - <<'PY'
import time, numpy as np
a=np.random.RandomState(0).rand(4096,4096).astype(np.float32)
t=time.time(); np.rint(a); print("rint 16.8M floats: %.3f s" % (time.time()-t))
PY
