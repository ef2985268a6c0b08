"""Developer probe: where the time of NetDesc.load_updated_parameters goes (device -> host copies, tensor loads, host packing + uploads)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np, torch
from cerberus_amd import _lib
from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs, make_state_dict
m = create_model(**default_model_kwargs()).train()
m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}, strict=True)
m.forward_train(torch.zeros((2, 64, 64, 3), dtype=torch.uint8).cuda())
dev = {k: v.cuda() for k, v in m._sd.items() if v.dtype == torch.float32}
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.time()
    for k, v in dev.items(): m._sd[k] = v.detach().cpu().clone()
    t1 = time.time()
    L, h = _lib.lib(), m._handle
    _lib.check(L.cerb_net_begin_reload(h))
    t2 = time.time()
    for k, v in m._sd.items():
        if v.dtype != torch.float32: continue
        a = np.ascontiguousarray(v.numpy()); shp = (C.c_int64 * a.ndim)(*a.shape)
        _lib.check(L.cerb_net_load_tensor(h, k.encode(), a.ctypes.data_as(C.c_void_p), shp, a.ndim))
    t3 = time.time()
    _lib.check(L.cerb_net_finalize(h))
    torch.cuda.synchronize(); t4 = time.time()
    print("to host %.3f s, begin_reload %.3f s, load_tensor x %d %.3f s, finalize (host packing + uploads) %.3f s" % (t1 - t0, t2 - t1, len(m._sd), t3 - t2, t4 - t3))
