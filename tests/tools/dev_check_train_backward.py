"""Developer check: every parameter's gradient statistics against the reference's train_step, worst tensors first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cerberus_amd.losses import PARAMSET_LOSS
from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs, make_state_dict
gold = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "train_loss.npz"))
m = create_model(**default_model_kwargs())
m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}, strict=True)
tiles = torch.from_numpy(gold["img"]).cuda()
keep = torch.from_numpy(gold["step/dropout_mask"].reshape(int(gold["N"]), 512)).cuda()
targets, flags = {}, {}
for j, h in enumerate(gold["heads"]):
    h = str(h); t = gold["target/" + h][..., 0]
    targets[h] = torch.from_numpy(t.reshape(t.shape[0]) if h == "Patch-Class" else t).cuda()
    flags[h] = torch.from_numpy(gold["has_target"][:, j].astype(np.float32)).cuda()
losses, grads = m.train_grads(tiles, targets, flags, PARAMSET_LOSS, keep)
rows = []
for k, (s_sum, s_abs, e0, em, e1) in zip([str(x) for x in gold["step/param_names"]], gold["step/grad_stats"]):
    if k.startswith("backbone.fc."): continue
    g = grads[k].double().flatten().cpu().numpy()
    floor_ = 1e-5 * g.size ** 0.5  # gradients that are mathematically zero (a bias in front of a BatchNorm) are rounding noise on both sides
    rows.append((abs(np.abs(g).sum() - s_abs) / max(s_abs, floor_), abs(g.sum() - s_sum) / max(s_abs, floor_), k, s_abs, np.abs(g).sum()))
rows.sort(reverse=True)
for r in rows[:int(sys.argv[1]) if len(sys.argv) > 1 else 25]:
    print("abs-sum rel err %.2e  sum err/abs %.2e  %-60s ref %.5g got %.5g" % r)
print("tensors with abs-sum error > 1e-3: %d of %d" % (sum(1 for r in rows if r[0] > 1e-3), len(rows)))
