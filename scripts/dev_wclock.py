"""Developer probe (library rebuilt with -DWPROF): shader clock while the Winograd kernel runs = s_memtime cycles / s_memrealtime
ticks (constant 100 MHz, hipDeviceAttributeWallClockRate) over the life of every workgroup of one forward's Winograd launches, and the
share of the launches' durations a workgroup is resident."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cerberus_amd import _lib
from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs
m = create_model(**default_model_kwargs())
t = torch.randint(0, 256, (32, 256, 256, 3), dtype=torch.uint8, device="cuda")
for _ in range(3): m.infer_tiles(t, 256)
torch.cuda.synchronize()
L = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * 8)()
m.profile(True)
L.cerb_dev_wprof(None, 1)
m.infer_tiles(t, 256); torch.cuda.synchronize()
L.cerb_dev_wprof(buf, 0)
cycles, n_wg, ticks = int(buf[3]), int(buf[4]), int(buf[7])
recs = [r for r in m.profile_records() if r[1].startswith("conv_wino")]
ms = sum(r[3] for r in recs)
print("Winograd launches %d, workgroups %d; shader clock while resident %.3f GHz (%.4g cycles / %.4g ticks of 10 ns)" % (len(recs), n_wg, cycles / ticks * 0.1, cycles, ticks))
print("mean residency of a workgroup: %.1f %% of its launch (sum of launch durations %.3f ms)" % (100.0 * ticks * 1e-5 / (ms * n_wg / len(recs)), ms))
