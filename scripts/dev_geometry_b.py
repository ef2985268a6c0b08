"""Developer probe: slide throughput with the reference's default geometry (448 in / 144 out) next to BASELINE's 256 / 256."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs
from cerberus_amd.wsi import WSIRunner, synth_slide
m = create_model(**default_model_kwargs())
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
slide = synth_slide(H, W, seed=2)
for win, out, batch in ((256, 256, 32), (448, 144, 10), (448, 144, 12), (448, 144, 20)):
    run = WSIRunner(m, (H, W), win, out, batch)
    if win == 448:  # a bounded sample: the first rows of patches
        run.n_patches = min(run.n_patches, 40 * batch)
    run.infer_band(slide, 0); torch.cuda.synchronize()
    t0 = time.time(); n = run.infer_band(slide, 0); torch.cuda.synchronize(); dt = time.time() - t0
    px_out = n * out * out
    print("win %d out %d batch %d: %d patches in %.3f s -> %.2f Mpx/s of slide (%.1f input Mpx/s, %.1f TFLOP/s)" % (
        win, out, batch, n, dt, px_out / dt / 1e6, n * win * win / dt / 1e6, n * win * win * 1.848e6 / dt / 1e12), flush=True)
