"""Developer instrumentation of conv_wino4p (build with -DP4_PROF=<workgroup> [-DP4_PROF_WAVE=<wave>]): per-step cycle stamps of one wave for
the last planar launch of a batch step (dec.3.1: Cin = 64, 4 chunks)."""
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
dt, step, n = bench.batch_loop(m, dev, 0, 3, 2, None, "nccl")
torch.cuda.synchronize()
L = _lib.lib()
buf = (C.c_ulonglong * (16 * 40))()
L.cerb_w4p_prof_read.argtypes = [C.c_void_p]
assert L.cerb_w4p_prof_read(buf) == 0
t = np.array(buf, dtype=np.uint64).reshape(16, 40).astype(np.int64)
tot = 0
for ch in range(4):
    row = t[ch, :36]
    nxt = t[ch + 1, 0] if ch < 3 else t[15, 0]
    d = np.diff(np.concatenate([row, [nxt]]))
    tot += d.sum()
    print("chunk %d: total %6d | " % (ch, d.sum()) + " ".join("%d" % v for v in d))
print("chunks %d cycles; output stage (after the next item's weight requests .. all stores issued) %d; store drain (vmcnt 0) %d" %
      (tot, t[15, 1] - t[15, 0], t[15, 2] - t[15, 1]))
