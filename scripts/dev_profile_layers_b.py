"""Developer tool: per-launch table of one forward in the reference's default geometry (448 in / 144 out)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10
m = create_model(**default_model_kwargs())
t = torch.randint(0, 256, (nb, 448, 448, 3), dtype=torch.uint8, device="cuda")
for _ in range(2): m.infer_tiles(t, 144)
torch.cuda.synchronize()
m.profile(True)
m.infer_tiles(t, 144)
torch.cuda.synchronize()
recs = m.profile_records()
tot = sum(r[3] for r in recs)
fam = {}
for name, kern, fl, ms in recs:
    if name.startswith("dec.") or name.startswith("head") or name in ("stem", "maxpool"):
        print("%-34s %-32s %9.1f %8.3f %8.1f" % (name, kern, fl / 1e9, ms, fl / ms / 1e9 if ms > 0 else 0))
    f = fam.setdefault(kern, [0, 0, 0]); f[0] += fl; f[1] += ms; f[2] += 1
print("total %.3f ms" % tot)
for k, (fl, ms, c) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print("%-34s n=%2d %8.3f ms (%4.1f%%) %7.1f TFLOP/s" % (k, c, ms, 100 * ms / tot, fl / ms / 1e9 if ms else 0))
