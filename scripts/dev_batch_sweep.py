"""Developer probe: forward time vs batch size (256^2 tiles, all heads)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs
m = create_model(**default_model_kwargs())
for nb in (1, 2, 4, 8, 16, 32, 48, 64, 96, 128):
    t = torch.randint(0, 256, (nb, 256, 256, 3), dtype=torch.uint8, device="cuda")
    for _ in range(3): m.infer_tiles(t, 256)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m.infer_tiles(t, 256)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("batch %2d: %7.2f ms  %6.1f Mpx/s  %6.1f TFLOP/s" % (nb, dt * 1e3, nb * 65536 / dt / 1e6, m.flops(nb, 256, 256) / dt / 1e12), flush=True)
