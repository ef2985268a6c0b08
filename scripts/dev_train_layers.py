"""Developer: per-launch records of one profiled training step (configs[4]: batch 16 x 448^2) for the families whose name contains argv[1].
    python scripts/dev_train_layers.py wgrad_wino4 [batch] [tile]"""
import sys

import torch

sys.path.insert(0, ".")
from cerberus_amd.losses import PARAMSET_LOSS  # noqa: E402
from cerberus_amd.net_desc import create_model  # noqa: E402
from cerberus_amd.train import Adam, train_step  # noqa: E402
from cerberus_amd.weights import default_model_kwargs, make_state_dict  # noqa: E402

pat = sys.argv[1] if len(sys.argv) > 1 else "wgrad"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
hw = int(sys.argv[3]) if len(sys.argv) > 3 else 448
dev = torch.device("cuda", 0)
kw = default_model_kwargs()
m = create_model(**kw)
m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}, strict=True)
g = torch.Generator(device=dev).manual_seed(0)
batch = {"img": torch.randint(0, 256, (n, hw, hw, 3), device=dev, generator=g, dtype=torch.uint8)}
heads = {d[3]: d[2] for d in m._decoders}
import numpy as np  # noqa: E402

has = np.empty((n, len(heads)), dtype=object)
for j, h in enumerate(heads):
    has[:, j] = h
batch["dummy_target"] = has
for h, c in heads.items():
    batch[h] = torch.randint(0, c, (n,), device=dev, generator=g).float() if h == "Patch-Class" else (
        (torch.rand((n, hw, hw, 1), device=dev, generator=g) < 0.3) * torch.randint(1, c, (n, hw, hw, 1), device=dev, generator=g)).float()
info = ({"net": {"desc": m, "optimizer": Adam(lr=1e-4), "extra_info": {"loss": PARAMSET_LOSS}}}, None)
for _ in range(2):
    train_step(batch, info)
m.profile(True)
train_step(batch, info)
torch.cuda.synchronize()
recs = m.profile_records()
m.profile(False)
tot = 0.0
for name, kern, work, ms in recs:
    if pat in kern:
        ex = work / (ms * 1e-3) / 1e12 * (0.25 if ("wino4" in kern) else 1.0)
        print("%-34s %-28s %8.3f ms  %7.2f TFLOP/s executed  %.3f" % (name[-34:], kern[:28], ms, ex, ex / 157.3))
        tot += ms
print("total %.3f ms" % tot)
