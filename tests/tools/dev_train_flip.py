"""debug (GPU): which elements of a parameter's first Adam step change sign between conv algorithms, and how small their gradients are"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs, make_state_dict
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from test_train_loss_gpu import PARAMSET_LOSS  # noqa
gold = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "train_loss.npz"), allow_pickle=True)
heads = [str(h) for h in gold["heads"]]
N = int(gold["N"])
dev = torch.device("cuda", 0)
targets, flags = {}, {}
for j, h in enumerate(heads):
    t = torch.from_numpy(gold["target/" + h]).float()
    targets[h] = (t.reshape(N) if h == "Patch-Class" else t.reshape(N, t.shape[1], t.shape[2])).to(dev)
    flags[h] = torch.from_numpy(gold["has_target"][:, j].astype(np.float32)).to(dev)
keep = torch.from_numpy(gold["step/dropout_mask"].reshape(N, 512)).cuda()
key = sys.argv[1] if len(sys.argv) > 1 else "decoder_head.Lumen.1.block.1.bn.weight"
def run(algo):
    m = create_model(**default_model_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(int(gold["weight_seed"])).items()}, strict=True)
    m.train(True)
    m.set_conv_algo(algo)
    losses, grads = m.train_grads(torch.from_numpy(gold["img"]).to(dev), targets, flags, PARAMSET_LOSS, keep, views=True)
    return {k: v.detach().clone().cpu().double().numpy() for k, v in grads.items() if not k.endswith(("running_mean", "running_var"))}
g1 = run(1)
g6 = run(6)
a, b = g1[key].ravel(), g6[key].ravel()
fl = np.nonzero(np.sign(a) != np.sign(b))[0]
print(key, "n", a.size, "max|g|", np.abs(a).max(), "flips", fl.tolist(), "values F2x2", a[fl], "F4x4", b[fl])
tot = 0
for k in g1:
    x, y = g1[k].ravel(), g6[k].ravel()
    nf = int((np.sign(x) != np.sign(y)).sum())
    tot += nf
    if nf and x.size <= 4096:
        f = np.nonzero(np.sign(x) != np.sign(y))[0]
        print("%-60s n %6d flips %3d  max|g| %.3e  largest flipped |g| %.3e  rel %.2e" % (k, x.size, nf, np.abs(x).max(), np.abs(x[f]).max(), np.abs(x[f]).max() / np.abs(x).max()))
print("total flips", tot)
