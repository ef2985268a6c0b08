"""Developer probe: host-resident slide -> inference, band uploaded up front vs chunk by chunk underneath the inference."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs
from cerberus_amd.wsi import SlabUploader, WSIRunner
m = create_model(**default_model_kwargs())
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
host = np.random.RandomState(0).randint(0, 256, (H, W, 3), dtype=np.uint8)
run = WSIRunner(m, (H, W), 256, 256, 32)
run.infer_band(torch.from_numpy(host[:512]).cuda().repeat(H // 512, 1, 1), 0); torch.cuda.synchronize()  # warm
for mode in ("up-front", "pipelined", "up-front", "pipelined"):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if mode == "up-front":
        slab = torch.from_numpy(host).cuda(); torch.cuda.synchronize(); t1 = time.perf_counter()
        run.infer_band(slab, 0)
    else:
        up = SlabUploader(host, 0, H); t1 = time.perf_counter()
        run.infer_band(up.slab, 0, ready=up.upload_until)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-9s: setup/upload %.3f s, total %.3f s -> %.1f Mpx/s including the host -> device copy" % (mode, t1 - t0, t2 - t0, H * W / (t2 - t0) / 1e6), flush=True)
