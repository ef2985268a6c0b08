"""Developer instrumentation of conv_wino4b (build with -DW4_PROF=<workgroup>): per-pair-step cycle stamps of one wave (last decoder conv)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from cerberus_amd import _lib
from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs, make_state_dict

dev = torch.device("cuda", 0)
m = create_model(**default_model_kwargs())
m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}, strict=True)
m._ensure_handle()
m.set_conv_algo(7)
dt, step, n = bench.batch_loop(m, dev, 0, 3, 2, None, "nccl")
torch.cuda.synchronize()
L = _lib.lib()
buf = (C.c_ulonglong * (16 * 40))()
L.cerb_w4_prof_read.argtypes = [C.c_void_p]
assert L.cerb_w4_prof_read(buf) == 0
t = np.array(buf, dtype=np.uint64).reshape(16, 40).astype(np.int64)
nch = 2
for ch in range(nch):
    row = t[ch, :18]
    nxt = t[ch + 1, 0] if ch + 1 < nch else t[15, 0]
    d = np.diff(np.concatenate([row, [nxt]]))
    print("chunk %d: total %d | " % (ch, d.sum()) + " ".join("%d" % v for v in d))
print("output stage stamps rel. to its start (bias, T+lds writes done, barrier, stores issued, barrier, vmcnt(0)):", [int(v - t[15, 0]) for v in t[15, 1:6]])
