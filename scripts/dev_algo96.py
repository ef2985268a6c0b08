import os, sys
sys.path.insert(0, "/root/repo")
import torch
from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs
nb = 96
m = create_model(**default_model_kwargs())
t = torch.randint(0, 256, (nb, 256, 256, 3), dtype=torch.uint8, device="cuda")
for algo in (6, 5, 7):
    m.set_conv_algo(algo)
    for _ in range(2): m.infer_tiles(t, 256)
    torch.cuda.synchronize()
    m.profile(True); m.infer_tiles(t, 256); torch.cuda.synchronize()
    recs = m.profile_records(); m.profile(False)
    print("algo", algo, "total %.3f ms" % sum(r[3] for r in recs))
    for name, kern, fl, ms in recs:
        if name.startswith("backbone.layer") and name.endswith(("1.conv1", "1.conv2")) or name.startswith("dec.0") or name.startswith("dec.1"):
            print("   %-28s %-34s %7.3f ms %6.1f TF" % (name, kern, ms, fl / ms / 1e9 if ms else 0))
