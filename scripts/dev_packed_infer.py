"""Developer: inference at the REFERENCE's default patch geometry (448-pixel input, 144-pixel kept window; infer/tile.py:43-106) with conv_wino4b's packed work items on / off.
    python scripts/dev_packed_infer.py [batch] [win] [out]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from cerberus_amd.net_desc import create_model  # noqa: E402
from cerberus_amd.weights import default_model_kwargs, make_state_dict  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
win = int(sys.argv[2]) if len(sys.argv) > 2 else 448
osz = int(sys.argv[3]) if len(sys.argv) > 3 else 144
kw = default_model_kwargs()
m = create_model(**kw)
m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}, strict=True)
tiles = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (n, win, win, 3)).astype(np.uint8)).cuda()
for mode in (True, False, True, False):
    m.set_packed_items(mode)
    for _ in range(3):
        m.infer_tiles(tiles, osz)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        m.infer_tiles(tiles, osz)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    m.profile(True)
    m.infer_tiles(tiles, osz)
    torch.cuda.synchronize()
    recs = m.profile_records()
    m.profile(False)
    w4b = sum(r[3] for r in recs if r[1].startswith("conv_wino4b"))
    print("packed_items=%d  batch %d x %d^2 -> %d^2: %.3f ms per forward = %.2f Mpx/s of input, %.2f Mpx/s kept; conv_wino4b launches %.3f ms (%d)" % (
        mode, n, win, osz, dt * 1e3, n * win * win / dt / 1e6, n * osz * osz / dt / 1e6, w4b, sum(r[1].startswith("conv_wino4b") for r in recs)))
