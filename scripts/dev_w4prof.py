"""Developer instrumentation of conv_wino4 (build with -DW4_PROF=<workgroup>): per-step cycle stamps of one wave for one layer."""
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
m.set_conv_algo(5)
dt, step, n = bench.batch_loop(m, dev, 0, 3, 2, None, "nccl")
torch.cuda.synchronize()
L = _lib.lib()
buf = (C.c_ulonglong * (16 * 40))()
L.cerb_w4_prof_read.argtypes = [C.c_void_p]
assert L.cerb_w4_prof_read(buf) == 0
t = np.array(buf, dtype=np.uint64).reshape(16, 40).astype(np.int64)
# the last launch of a step that used the kernel is the last decoder conv (dec.3.1: Cin = 64, 4 chunks)
for ch in range(4):
    row = t[ch, :36]
    nxt = t[ch + 1, 0] if ch < 3 else t[4, 39]
    d = np.diff(np.concatenate([row, [nxt]]))
    print("chunk %d: total %d | " % (ch, d.sum()) + " ".join("%d" % v for v in d))
print("output stage + next item start:", t[4, 39] - t[3, 35])
st = t[15, :9]
print("output stage stamps (start, tb0 begin, tb0 T done, -, tb1 begin, tb1 T done, -, stores issued, vmcnt(0)):", [int(v - t[3, 35]) for v in st])
