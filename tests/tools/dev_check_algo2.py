"""Developer check: conv_algo 2 (Winograd with bf16x3 products) against the oracle and against algo 1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs, make_state_dict
from oracle import net_ref
kw = default_model_kwargs()
sd = {k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}
m = create_model(**kw); m.load_state_dict(sd, strict=True)
tiles = np.random.RandomState(11).randint(0, 256, (2, 256, 256, 3)).astype(np.uint8)
x = torch.from_numpy(tiles).float().permute(0, 3, 1, 2).contiguous()
ref = net_ref.net_forward(sd, x, kw["decoder_kwargs"], kw["considered_tasks"])
for algo in (1, 2):
    m.set_conv_algo(algo)
    out = m(x.cuda())
    print("algo %d logits max err: %s" % (algo, {k: "%.2e" % float((v.cpu() - ref[k]).abs().max()) for k, v in out.items()}), flush=True)
    t = torch.randint(0, 256, (32, 256, 256, 3), dtype=torch.uint8, device="cuda")
    for _ in range(3): m.infer_tiles(t, 256)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m.infer_tiles(t, 256)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("algo %d batch 32: %.2f ms  %.1f Mpx/s" % (algo, dt * 1e3, 32 * 65536 / dt / 1e6), flush=True)
m.profile(True); m.infer_tiles(t, 256); torch.cuda.synchronize()
fam = {}
for name, kern, fl, ms in m.profile_records():
    f = fam.setdefault(kern, [0, 0, 0]); f[0] += fl; f[1] += ms; f[2] += 1
for k, (fl, ms, c) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:6]:
    print("%-34s n=%2d %8.3f ms %7.1f TFLOP/s" % (k, c, ms, fl / ms / 1e9 if ms else 0))
