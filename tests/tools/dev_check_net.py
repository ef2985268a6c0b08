"""Developer diagnostic: per-stage max-abs error of the HIP forward vs the CPU oracle (not a test)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs, make_state_dict
from oracle import net_ref

hw = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
kw = default_model_kwargs()
sd_np = make_state_dict(0)
sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
m = create_model(**kw)
m.load_state_dict(sd, strict=True)
tiles = np.random.RandomState(1).randint(0, 256, (n, hw, hw, 3)).astype(np.uint8)
x = torch.from_numpy(tiles).float().permute(0, 3, 1, 2).contiguous()
t0 = time.time()
ref, feats, bottom = net_ref.net_forward(sd, x, kw["decoder_kwargs"], kw["considered_tasks"], return_feats=True)
print("oracle forward %.2fs" % (time.time() - t0))
tt = torch.from_numpy(tiles).cuda()
f = m.encoder_features(tt)
torch.cuda.synchronize()
names = ["x0", "x1", "x2", "x3", "cm", "x4"]
refs = feats[:4] + [feats[4], bottom]
for nm, a, b in zip(names, f, refs):
    a = a.cpu().permute(0, 3, 1, 2)
    print("%-4s shape %-22s max|ref| %.3f  maxabs err %.3e" % (nm, tuple(a.shape), b.abs().max().item(), (a - b).abs().max().item()))
out = m(tt)
torch.cuda.synchronize()
for k, v in out.items():
    r = ref[k]
    print("%-12s logits max|ref| %.3f  maxabs err %.3e" % (k, r.abs().max().item(), (v.cpu() - r).abs().max().item()))
o = m.infer_tiles(tt, hw)
oo = net_ref.infer_step(sd, tiles, hw, kw["considered_tasks"], kw["decoder_kwargs"])
for k, v in o.items():
    r = np.stack([oo[i][k] for i in range(n)])
    v = v.cpu().numpy()
    if v.dtype == np.float32:
        print("%-12s out maxabs err %.3e" % (k, np.abs(v - r).max()))
    else:
        print("%-12s out mismatch frac %.3e" % (k, (v != r).mean()))
# timing
for nb in (8, 32):
    tb = torch.randint(0, 256, (nb, 256, 256, 3), dtype=torch.uint8, device="cuda")
    m.infer_tiles(tb, 256); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3): m.infer_tiles(tb, 256)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 3
    fl = m.flops(nb, 256, 256)
    print("batch %d: %.2f ms  %.1f TFLOP/s  %.1f Mpx/s" % (nb, dt * 1e3, fl / dt / 1e12, nb * 65536 / dt / 1e6))
