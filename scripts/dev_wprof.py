"""Developer probe: per-phase cycle shares of conv_wino (library rebuilt with -DWPROF by scripts/dev_wprof.sh)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cerberus_amd import _lib
from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs
m = create_model(**default_model_kwargs())
t = torch.randint(0, 256, (32, 256, 256, 3), dtype=torch.uint8, device="cuda")
m.infer_tiles(t, 256); torch.cuda.synchronize()
L = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * 8)()
m.profile(True)
L.cerb_dev_wprof(None, 1)
m.infer_tiles(t, 256); torch.cuda.synchronize()
L.cerb_dev_wprof(buf, 0)
b, mm, e, tot, n, brd, pro = [int(buf[i]) for i in range(7)]
print("all wino launches of one forward: workgroups %d; wave-0 cycles: boundary(barrier+V write+barrier) %.1f%%  MFMA phase %.1f%%  output stage %.1f%%  border-item mask+transform %.1f%%  prologue %.1f%%  rest (item setup, timers) %.1f%%" % (
    n, 100.0 * b / tot, 100.0 * mm / tot, 100.0 * e / tot, 100.0 * brd / tot, 100.0 * pro / tot, 100.0 * (tot - b - mm - e - brd - pro) / tot))
