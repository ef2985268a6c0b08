import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cerberus_amd.postproc import postproc_device
from cerberus_amd import synth_maps as synth
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
m = torch.from_numpy(synth.nuclei_maps(H, W, 7, 1000.0, noise=0.02)).cuda()
g = torch.from_numpy(synth.blob_maps(H, W, 9, int(30 * H * W / 1e6), 14.0, 60.0, rim=4.0, sharp=1.0, noise=0.02, holes=0.3)).cuda()
for t, x in (("Nuclei", m), ("Gland", g), ("Lumen", g)):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        lab, info = postproc_device(x, t)
        torch.cuda.synchronize(); dt = time.time() - t0
    print("%s %dx%d: %.2f ms (%.0f Mpx/s) n_inst %d" % (t, H, W, dt * 1e3, H * W / dt / 1e6, int(info["n_inst"])), flush=True)
