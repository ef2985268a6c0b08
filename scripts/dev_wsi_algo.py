"""Developer probe: slide inference throughput (256 / 256, batch 96) under the F(4x4) kernel choices."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs
from cerberus_amd.wsi import WSIRunner, synth_slide
m = create_model(**default_model_kwargs())
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 96
slide = synth_slide(H, W, seed=2)
run = WSIRunner(m, (H, W), 256, 256, batch)
for algo in (6, 7, 5, 6, 7):
    m._ensure_handle(); m.set_conv_algo(algo)
    run.infer_band(slide, 0); torch.cuda.synchronize()
    t0 = time.time(); n = run.infer_band(slide, 0); torch.cuda.synchronize(); dt = time.time() - t0
    print("algo %d batch %d: %d patches in %.3f s -> %.2f Mpx/s" % (algo, batch, n, dt, n * 65536 / dt / 1e6), flush=True)
