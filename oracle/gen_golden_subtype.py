"""Generate tests/golden/train_subtype.npz: the REFERENCE's own `train_step` (models/run_desc.py:25-230) in its sub-typing configuration
(`subtype_nuclei=True`: models/net_desc.py:105-142 `_freeze_weight` + the decoder gating of :160-170), run on CPU in this container.

Run (py3.10 + torch):  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_subtype.py

What the reference does in that mode, and what the fixture lets the tests check:
  * backbone, conv_map, Patch-Class, every INST decoder / head and the unselected TYPE decoder / head are frozen: requires_grad False
    and their BatchNorm layers in eval mode -- the forward uses their RUNNING statistics and leaves them untouched;
  * the selected decoder ("Nuclei#TYPE") and its head stay in training mode (batch statistics, running statistics updated) and are
    the only parameters Adam moves -- of the decoder only the last block's, because the reference runs "#TYPE" decoders under
    torch.set_grad_enabled(False) ("Nuclei#TYPE" is never in train_decoder_list) and its conv layers re-enable autograd inside themselves.
Stored: inputs (image, targets, flags, dropout mask), the train-mode logits of every head, the reported losses, per-parameter update
statistics (|after - before| sum, first / middle / last element) and per-buffer running-statistics change.  Same batch recipe as
gen_golden_train_loss.py; the Nuclei-TYPE loss weight is 1 (paramset.yml switches that head off, which would leave nothing to train)."""
import os
import sys
from collections import OrderedDict
from unittest.mock import MagicMock

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True
for m in ["cv2", "skimage", "skimage.filters", "skimage.morphology", "termcolor", "matplotlib", "matplotlib.pyplot", "tensorboardX", "imgaug",
          "imgaug.augmenters"]:
    if m not in sys.modules:
        try:
            __import__(m)
        except Exception:
            sys.modules[m] = MagicMock()
_orig_to = torch.Tensor.to


def _to(self, *a, **k):
    if a and a[0] == "cuda":
        return self
    return _orig_to(self, *a, **k)


torch.Tensor.to = _to

from models.net_desc import create_model  # noqa: E402  (reference)
from models.run_desc import train_step  # noqa: E402  (reference)

from cerberus_amd.weights import default_model_kwargs, make_state_dict  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    rs = np.random.RandomState(23)
    kw = default_model_kwargs()
    kw["subtype_nuclei"] = True
    model = create_model(**kw)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}, strict=True)
    net = torch.nn.DataParallel(model)
    opt = torch.optim.Adam(net.parameters(), lr=1.0e-3, betas=(0.9, 0.999))
    loss_kwargs = yaml.full_load(open("/root/reference/models/paramset.yml"))["loss_kwargs"]
    loss_kwargs["loss_info"]["Nuclei-TYPE"]["weight"] = 1.0
    N, H = 3, 64
    heads = OrderedDict([("Lumen-INST", 3), ("Gland-INST", 3), ("Nuclei-INST", 3), ("Nuclei-TYPE", 7), ("Gland-TYPE", 3), ("Patch-Class", 9)])
    batch = {"img": torch.from_numpy(rs.randint(0, 256, (N, H, H, 3)).astype(np.uint8))}
    targets = {}
    for h, c in heads.items():
        if h == "Patch-Class":
            t = rs.randint(0, c, (N, 1, 1, 1))
        else:
            t = (rs.rand(N, H, H, 1) < 0.35) * rs.randint(1, c, (N, H, H, 1))
            t[:, :8] = 0
        targets[h] = t.astype(np.float32)
        batch[h] = torch.from_numpy(targets[h])
    has = np.full((N, len(heads)), None, dtype=object)
    for j, h in enumerate(heads):
        for n in range(N):
            if n == 1 and h.startswith("Gland"):
                continue
            has[n, j] = h
    batch["dummy_target"] = has
    captured, drop = {}, {}

    def hook(name):
        def f(mod, inp, out):
            captured[name] = out.detach().numpy().copy()
        return f

    for dec, hd in model.output_head.items():
        for clf, mod in hd.items():
            mod.register_forward_hook(hook(dec.split("#")[0] + "-" + clf))
    model.decoder_head["Patch-Class"].register_forward_hook(hook("Patch-Class"))

    def drop_hook(mod, inp, out):
        drop["mask"] = (out != 0).detach().numpy() | (inp[0] == 0).detach().numpy()
        drop["training"] = bool(mod.training)

    model.decoder_head["Patch-Class"].dropout.register_forward_hook(drop_hook)
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    run_info = ({"net": {"desc": net, "optimizer": opt, "extra_info": {"loss": loss_kwargs}}}, None)
    res = train_step(dict(batch), run_info)
    store = {"N": N, "H": H, "img": batch["img"].numpy(), "weight_seed": 0, "heads": np.array(list(heads.keys())), "n_classes": np.array(list(heads.values())),
             "has_target": np.array([[x is not None for x in row] for row in has]), "dropout_mask": drop["mask"], "dropout_training": np.array(drop["training"]),
             "overall_loss": np.float64(res["EMA"]["overall_loss"]),
             "loss_weight": np.array([loss_kwargs["loss_info"][h]["weight"] for h in heads], np.float64)}
    for h in heads:
        store["target/" + h] = targets[h]
        store["logits/" + h] = captured[h]
        store["loss/" + h] = np.float64(res["EMA"]["%s_loss" % h])
        print("%-12s loss %.6f" % (h, store["loss/" + h]))
    names, pstat, moved = [], [], []
    after = model.state_dict()
    for k, prm in model.named_parameters():
        d64 = (prm.detach().double().flatten() - before[k].double().flatten())
        names.append(k)
        pstat.append([prm.detach().double().sum().item(), d64.abs().sum().item(), d64[0].item(), d64[d64.numel() // 2].item(), d64[-1].item()])
        if d64.abs().sum().item() > 0:
            moved.append(k)
    store["param_names"] = np.array(names)
    store["update_stats"] = np.array(pstat)
    store["moved"] = np.array(moved)
    bn_names, bn_stat = [], []
    for k, v in after.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            bn_names.append(k)
            bn_stat.append([v.double().sum().item(), (v.double() - before[k].double()).abs().sum().item(), v.double().flatten()[0].item()])
    store["bn_names"] = np.array(bn_names)
    store["bn_stats"] = np.array(bn_stat)
    tracked = [k for k, v in after.items() if k.endswith("num_batches_tracked") and int(v) != int(before[k])]
    store["tracked_moved"] = np.array(tracked)
    print("overall %.6f; %d of %d parameters moved: %s ..." % (store["overall_loss"], len(moved), len(names), moved[:3]))
    print("running statistics changed in %d of %d buffers; num_batches_tracked advanced in %d; dropout module training=%s" %
          (int((np.array(bn_stat)[:, 1] > 0).sum()), len(bn_names), len(tracked), drop["training"]))
    print("moved prefixes:", sorted(set(k.split(".block")[0] if "block" in k else k.rsplit(".", 1)[0] for k in moved)))
    path = os.path.join(ROOT, "tests", "golden", "train_subtype.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
