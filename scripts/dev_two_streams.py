"""Developer probe: K handles on K streams, batches alternating -- do the tails of one stream's launches fill with the other's?
    python scripts/dev_two_streams.py <tiles per batch> <streams>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs, make_state_dict
sd = {k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}
ms = []
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for i in range(K):
    m = create_model(**default_model_kwargs()); m.load_state_dict(sd, strict=True); ms.append(m)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 48
tiles = [torch.randint(0, 256, (nb, 256, 256, 3), dtype=torch.uint8, device="cuda") for _ in range(K)]
streams = [torch.cuda.Stream() for _ in range(K)]
def run(two, reps=12):
    for r in range(reps):
        i = r % K
        if two:
            with torch.cuda.stream(streams[i]):
                ms[i].infer_tiles(tiles[i], 256)
        else:
            ms[i].infer_tiles(tiles[i], 256)
    torch.cuda.synchronize()
for two in (False, True, False, True):
    run(two, 4)
    t0 = time.perf_counter(); run(two, 12); dt = time.perf_counter() - t0
    print("%d streams" % K if two else "one stream ", "batch %d: %.2f ms per batch, %.1f Mpx/s" % (nb, dt / 12 * 1e3, 12 * nb * 65536 / dt / 1e6), flush=True)
