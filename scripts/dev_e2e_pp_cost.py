"""Developer probe: cost of post-processing on the network's OWN output canvas (degenerate random-weight maps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs
from cerberus_amd.wsi import WSIRunner, synth_slide
m = create_model(**default_model_kwargs())
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
slide = synth_slide(H, W, seed=2)
run = WSIRunner(m, (H, W), 256, 256, 32)
run.infer_band(slide, 0); torch.cuda.synchronize()
t0 = time.time(); run.infer_band(slide, 0); torch.cuda.synchronize(); t1 = time.time() - t0
print("infer %dx%d: %.1f ms  %.1f Mpx/s" % (H, W, t1 * 1e3, H * W / t1 / 1e6), flush=True)
full = run.gather_to_root()
for k, v in full.items():
    if v.dim() == 3: print(k, "inner>0.5 frac %.3f  cnt>0.5 frac %.3f" % ((v[..., 0] > 0.5).float().mean().item(), (v[..., 1] > 0.5).float().mean().item()), flush=True)
from cerberus_amd.postproc import postproc_device
for t in ("Nuclei", "Gland", "Lumen"):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        lab, info = postproc_device(full[t + "-INST"], t)
        torch.cuda.synchronize(); dt = time.time() - t0
    print("%s postproc: %.2f ms (%.1f Mpx/s) n_inst %d amb %d" % (t, dt * 1e3, H * W / dt / 1e6, int(info["n_inst"]), int(info["n_ambiguous"])), flush=True)
