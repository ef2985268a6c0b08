"""Developer probe (library rebuilt with -DWPROF): when do the 512 persistent workgroups of every Winograd launch start and finish?
Per launch: window = first start .. last end (100 MHz ticks), share of the window the average workgroup is resident, spread of the
starts and of the ends."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cerberus_amd import _lib
from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs
m = create_model(**default_model_kwargs())
t = torch.randint(0, 256, (32, 256, 256, 3), dtype=torch.uint8, device="cuda")
for _ in range(3): m.infer_tiles(t, 256)
torch.cuda.synchronize()
L = C.CDLL(_lib.LIB_PATH)
L.cerb_dev_wlog(None, None, 1)
m.infer_tiles(t, 256); torch.cuda.synchronize()
buf = (C.c_ulonglong * (4 * 32768))(); n = C.c_uint(0)
L.cerb_dev_wlog(buf, C.byref(n), 0)
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4)[: n.value].astype(np.int64)
order = np.argsort(a[:, 0], kind="stable"); a = a[order]
# launches are separated in time: split where a start comes after every earlier end
launches, cur_end, s0 = [], a[0, 1], 0
for i in range(1, len(a)):
    if a[i, 0] > cur_end:
        launches.append(a[s0:i]); s0 = i; cur_end = a[i, 1]
    else:
        cur_end = max(cur_end, a[i, 1])
launches.append(a[s0:])
tot_win = tot_res = 0.0
print("%3s %5s %4s %4s %9s %9s %9s %9s %9s" % ("#", "wgs", "H", "Cin", "window us", "resident%", "start p99", "end p1", "gap us"))
prev_end = None
for k, l in enumerate(launches):
    w0, w1 = l[:, 0].min(), l[:, 1].max()
    win = (w1 - w0) / 100.0
    res = (l[:, 1] - l[:, 0]).sum() / 100.0 / len(l)
    gap = (w0 - prev_end) / 100.0 if prev_end is not None else 0.0
    prev_end = w1
    tot_win += win; tot_res += res
    print("%3d %5d %4d %4d %9.1f %9.1f %9.1f %9.1f %9.1f" % (k, len(l), l[0, 2] >> 32, l[0, 2] & 0xffffffff, win, 100.0 * res / win,
          (np.percentile(l[:, 0], 99) - w0) / 100.0, (w1 - np.percentile(l[:, 1], 1)) / 100.0, gap))
print("sum of windows %.1f us, mean residency %.1f%%" % (tot_win, 100.0 * tot_res / tot_win))
# which workgroups finish early?  blockIdx -> XCD = b % 8, dispatch order inside the XCD = b // 8 (first 32 take the first slot of a CU?)
for k in (0, 13, 24, 35):
    l = launches[k]
    w0, w1 = l[:, 0].min(), l[:, 1].max()
    j = l[:, 3] // 8
    e = (l[:, 1] - w0) / 100.0
    first, second = e[j < 32], e[j >= 32]
    print("launch %d: window %.1f us; end time of workgroups dispatched 1st..32nd in their XCD: mean %.1f (min %.1f max %.1f); 33rd..64th: mean %.1f (min %.1f max %.1f)" % (
        k, (w1 - w0) / 100.0, first.mean(), first.min(), first.max(), second.mean(), second.min(), second.max()))
    h = np.histogram(e / e.max(), bins=10, range=(0, 1))[0]
    print("   histogram of end time / window (10 bins):", h.tolist())
