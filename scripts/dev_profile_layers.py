"""Developer tool: per-launch table (HIP events) of one forward at batch 32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = create_model(**default_model_kwargs())
t = torch.randint(0, 256, (nb, 256, 256, 3), dtype=torch.uint8, device="cuda")
for _ in range(2): m.infer_tiles(t, 256)
torch.cuda.synchronize()
m.profile(True)
m.infer_tiles(t, 256)
torch.cuda.synchronize()
recs = m.profile_records()
tot = sum(r[3] for r in recs)
print("%-34s %-32s %9s %8s %8s" % ("layer", "kernel", "GFLOP", "ms", "TFLOP/s"))
for name, kern, fl, ms in recs:
    print("%-34s %-32s %9.1f %8.3f %8.1f" % (name, kern, fl / 1e9, ms, fl / ms / 1e9 if ms > 0 else 0))
print("total %.3f ms  -> %.1f TFLOP/s" % (tot, sum(r[2] for r in recs) / tot / 1e9))
fam = {}
for name, kern, fl, ms in recs:
    f = fam.setdefault(kern, [0, 0, 0]); f[0] += fl; f[1] += ms; f[2] += 1
for k, (fl, ms, c) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print("%-34s n=%2d %8.3f ms (%4.1f%%) %7.1f TFLOP/s" % (k, c, ms, 100 * ms / tot, fl / ms / 1e9 if ms else 0))
