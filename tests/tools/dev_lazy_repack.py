"""debug (GPU): a training handle that never used its F(2x2) packing, then switches to conv_algo 1 after an optimiser step, must compute with
freshly packed weights: its second step == the step of a handle built from the state dict after step 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cerberus_amd.net_desc import create_model
from cerberus_amd.train import Adam, train_step
from cerberus_amd.weights import default_model_kwargs, make_state_dict
from test_train_loss_gpu import PARAMSET_LOSS
gold = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "train_loss.npz"), allow_pickle=True)
heads = [str(h) for h in gold["heads"]]
N = int(gold["N"])
has = np.full((N, len(heads)), None, dtype=object)
for j, h in enumerate(heads):
    for n in range(N):
        if gold["has_target"][n, j]:
            has[n, j] = h
batch = {"img": torch.from_numpy(gold["img"]), "dummy_target": has}
for h in heads:
    batch[h] = torch.from_numpy(gold["target/" + h])
keep = torch.from_numpy(gold["step/dropout_mask"].reshape(N, 512)).cuda()
def mk(sd):
    m = create_model(**default_model_kwargs()); m.load_state_dict(sd, strict=True); m.train(True); return m
sd0 = {k: torch.from_numpy(v) for k, v in make_state_dict(int(gold["weight_seed"])).items()}
a = mk(sd0); opt = Adam(lr=1e-3)
a.set_conv_algo(6)
train_step(batch, ({"net": {"desc": a, "optimizer": opt, "extra_info": {"loss": PARAMSET_LOSS}}}, None), dropout_keep=keep)
sd1 = {k: v.clone() for k, v in a.state_dict().items()}
a.set_conv_algo(1)
r2 = train_step(batch, ({"net": {"desc": a, "optimizer": opt, "extra_info": {"loss": PARAMSET_LOSS}}}, None), dropout_keep=keep)
c = mk(sd1); c.set_conv_algo(1)
lc, gc = c.train_grads(torch.from_numpy(gold["img"]).cuda(), *[None] * 0) if False else (None, None)
# compare the LOSSES of the second step (forward on re-packed vs freshly packed weights): bitwise
import copy
optc = Adam(lr=1e-3)
rc = train_step(batch, ({"net": {"desc": c, "optimizer": optc, "extra_info": {"loss": PARAMSET_LOSS}}}, None), dropout_keep=keep)
for k in r2["EMA"]:
    print(k, r2["EMA"][k], rc["EMA"][k], "OK" if r2["EMA"][k] == rc["EMA"][k] else "DIFF")
