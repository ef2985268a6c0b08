"""Developer timing: why does `Preparing Input Output Placement` take 6.5 s for the second 3.2-Gpx slide of a directory and 2 s for the first?"""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cerberus_amd.tile import InferManager  # noqa: E402
from cerberus_amd.weights import DEFAULT_REQ_TARGET_CODE, default_model_kwargs  # noqa: E402
from cerberus_amd.wsi import WSIRunner  # noqa: E402

mgr = InferManager(checkpoint_path=None, decoder_dict=dict(DEFAULT_REQ_TARGET_CODE), model_args=default_model_kwargs())
H, W = 49152, 65536
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run = WSIRunner(mgr.net, (H, W), 256, 256, 64)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    lab = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    ws = torch.empty(38 << 30, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    del run, lab, ws
    gc.collect()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("rep %d: WSIRunner %.2f s, labels + workspace %.2f s, release %.2f s; reserved %.1f GB allocated %.1f GB" % (
        rep, t1 - t0, t2 - t1, t3 - t2, torch.cuda.memory_reserved() / 1e9, torch.cuda.memory_allocated() / 1e9))
