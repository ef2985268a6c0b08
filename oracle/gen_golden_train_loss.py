"""Generate tests/golden/train_loss.npz by running the REFERENCE's own `train_step` (models/run_desc.py:25-230) on CPU in this container.

Run (py3.10 + torch):  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_train_loss.py
Needs /root/reference (read-only); never runs on the GPU box.  What is captured is DATA: for every output head the logits the
reference's network produced in train mode (forward hook on the head module), the targets / per-sample target flags fed to it, the
loss value train_step reports for the head (result_dict["EMA"]["<head>_loss"]), the overall loss, and d(overall loss)/d(logits) that
`all_loss.backward()` left on the logits (retain_grad).  Loss options are the reference's models/paramset.yml `loss_kwargs`.

Inert stubs / shims as in gen_golden_net.py: cv2 / skimage / termcolor ... are only reached by imports; `.to("cuda")` is neutralised
because this container has no GPU; torch.nn.DataParallel without devices simply calls the module (train_step needs `.module`)."""
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


def run_case(prefix, nuclei_type_weight, store, weight_maps=False):
    torch.manual_seed(0)
    torch.set_num_threads(8)
    rs = np.random.RandomState(11)
    kw = default_model_kwargs()
    model = create_model(**kw)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}, strict=True)
    net = torch.nn.DataParallel(model)
    opt = torch.optim.Adam(net.parameters(), lr=1.0e-3, betas=(0.9, 0.999))
    loss_kwargs = yaml.full_load(open("/root/reference/models/paramset.yml"))["loss_kwargs"]
    if nuclei_type_weight is not None:  # paramset.yml switches this head off (weight 0); the second case exercises its class weights
        loss_kwargs["loss_info"]["Nuclei-TYPE"]["weight"] = nuclei_type_weight
    N, H = 3, 64
    heads = OrderedDict([("Lumen-INST", 3), ("Gland-INST", 3), ("Nuclei-INST", 3), ("Nuclei-TYPE", 7), ("Gland-TYPE", 3), ("Patch-Class", 9)])
    batch = {"img": torch.from_numpy(rs.randint(0, 256, (N, H, H, 3)).astype(np.uint8))}
    targets = {}
    for h, c in heads.items():
        if h == "Patch-Class":
            t = rs.randint(0, c, (N, 1, 1, 1))
        else:  # blobby class maps: mostly background with patches of the positive classes
            t = (rs.rand(N, H, H, 1) < 0.35) * rs.randint(1, c, (N, H, H, 1))
            t[:, :8] = 0
        targets[h] = t.astype(np.float32)
        batch[h] = torch.from_numpy(targets[h])
    wmap_heads = ("Gland-INST", "Nuclei-INST") if weight_maps else ()
    for h in wmap_heads:  # loader/targets.py:55-57: 1 + w0 exp(-d^2 / 2) outside the instances, 1 inside -- here any positive map serves
        wm = (1.0 + 4.0 * rs.rand(N, H, H, 1) * (targets[h] == 0)).astype(np.float32)
        batch[h + "#WEIGHT-MAP"] = torch.from_numpy(wm)
        store["wmap/weight_map/" + h] = wm
    # which sample carries which target: sample 1 has no gland annotation, sample 2 no nuclei types (dummy targets there)
    has = np.full((N, len(heads)), None, dtype=object)
    for j, h in enumerate(heads):
        for n in range(N):
            if (n == 1 and h.startswith("Gland")) or (n == 2 and h == "Nuclei-TYPE"):
                continue
            has[n, j] = h
    batch["dummy_target"] = has
    captured = {}

    def hook(name):
        def f(mod, inp, out):
            out.retain_grad()
            captured[name] = out
        return f

    for dec, hd in model.output_head.items():
        for clf, mod in hd.items():
            mod.register_forward_hook(hook(dec.split("#")[0] + "-" + clf))
    model.decoder_head["Patch-Class"].register_forward_hook(hook("Patch-Class"))
    drop = {}

    def drop_hook(mod, inp, out):  # nn.Dropout(p=0.3) of the Patch-Class branch (models/net_desc.py:70): keep mask of this step
        drop["mask"] = (out != 0).detach().numpy() | (inp[0] == 0).detach().numpy()

    model.decoder_head["Patch-Class"].dropout.register_forward_hook(drop_hook)
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    run_info = ({"net": {"desc": net, "optimizer": opt, "extra_info": {"loss": loss_kwargs}}}, None)
    res = train_step(dict(batch), run_info)
    store.update({"N": N, "H": H, "img": batch["img"].numpy(), "weight_seed": 0, "heads": np.array(list(heads.keys())), "n_classes": np.array(list(heads.values())),
                  "has_target": np.array([[x is not None for x in row] for row in has])})
    store[prefix + "overall_loss"] = np.float64(res["EMA"]["overall_loss"])
    for h in heads:
        lg = captured[h]
        if "logits/" + h in store:  # both cases start from the same seeds: the forward is identical, stored once
            assert np.array_equal(store["logits/" + h], lg.detach().numpy())
        store["logits/" + h] = lg.detach().numpy()                          # NCHW, as the reference's forward returns them
        g = lg.grad.numpy() if lg.grad is not None else np.zeros_like(lg.detach().numpy())
        if prefix == "wmap/":  # only the heads with a weight map differ from the paramset case
            if h in wmap_heads:
                store[prefix + "dlogits/" + h] = g
                assert not np.array_equal(store["paramset/dlogits/" + h], g)
            else:
                assert np.array_equal(store["paramset/dlogits/" + h], g), h
        elif prefix == "paramset/" or h == "Nuclei-TYPE":                   # the other heads' gradients do not depend on the case
            store[prefix + "dlogits/" + h] = g
        else:
            assert np.array_equal(store["paramset/dlogits/" + h], g)
        store["target/" + h] = targets[h]                                   # NHWC with one channel, float class ids (same in both cases)
        store[prefix + "loss/" + h] = np.float64(res["EMA"]["%s_loss" % h])
        print("%s%-12s logits %-18s loss %.6f  |dlogits| max %.3e" % (prefix, h, tuple(lg.shape), store[prefix + "loss/" + h],
                                                                      np.abs(g).max()))
    print(prefix, "overall", store[prefix + "overall_loss"])
    if prefix == "paramset/":  # the whole step, for the round that builds the backward pass: gradients, Adam update, BN statistics
        store["step/dropout_mask"] = drop["mask"]
        names, gstat, pstat = [], [], []
        for k, prm in model.named_parameters():
            g = prm.grad if prm.grad is not None else torch.zeros_like(prm)
            g64, p64 = g.double().flatten(), prm.detach().double().flatten()
            names.append(k)
            gstat.append([g64.sum().item(), g64.abs().sum().item(), g64[0].item(), g64[g64.numel() // 2].item(), g64[-1].item()])
            d64 = (prm.detach().double().flatten() - before[k].double().flatten())
            pstat.append([p64.sum().item(), d64.abs().sum().item(), d64[0].item(), d64[d64.numel() // 2].item(), d64[-1].item()])
        # FULL gradient tensors (fp32) for at least one layer per backward kernel family, compared element by element on the GPU:
        #   3x3 stride-1 weight gradient + Winograd data gradient upstream of it (backbone, decoder), 3x3 stride 2, 1x1 stride 2,
        #   plain 1x1 (conv_map), the 7x7 stem, the heads' two pointwise layers, BatchNorm gamma / beta, conv biases, Patch-Class
        full = ["backbone.conv1.weight", "backbone.bn1.weight", "backbone.bn1.bias", "backbone.layer1.0.conv1.weight", "backbone.layer1.0.bn1.weight",
                "backbone.layer1.0.bn1.bias", "backbone.layer1.2.conv2.weight", "backbone.layer2.0.conv1.weight", "backbone.layer2.0.downsample.0.weight",
                "backbone.layer2.0.downsample.1.weight", "backbone.layer3.0.conv1.weight", "backbone.layer4.0.downsample.0.weight", "conv_map.weight",
                "decoder_head.Nuclei.0.block.0.conv.weight", "decoder_head.Nuclei.3.block.0.conv.weight", "decoder_head.Nuclei.3.block.1.conv.weight",
                "decoder_head.Nuclei.3.block.1.conv.bias", "decoder_head.Nuclei.3.block.1.bn.weight", "decoder_head.Nuclei.3.block.1.bn.bias",
                "decoder_head.Lumen.2.block.0.conv.weight", "decoder_head.Gland#TYPE.3.block.1.conv.weight", "decoder_head.Gland#TYPE.3.block.1.bn.weight",
                "decoder_head.Gland#TYPE.3.block.1.bn.bias", "decoder_head.Gland#TYPE.3.block.0.conv.weight", "decoder_head.Nuclei#TYPE.3.block.1.conv.weight",
                "decoder_head.Gland.3.block.1.conv.weight", "output_head.Gland#TYPE.TYPE.x.0.block.0.conv.weight", "output_head.Nuclei.INST.x.0.block.0.conv.weight",
                "output_head.Nuclei.INST.x.0.block.0.bn.weight", "output_head.Nuclei.INST.x.1.conv.weight", "output_head.Nuclei.INST.x.1.conv.bias",
                "output_head.Gland#TYPE.TYPE.x.1.conv.weight", "decoder_head.Patch-Class.conv1.weight", "decoder_head.Patch-Class.conv2.weight",
                "decoder_head.Patch-Class.bn1.weight"]
        prm = dict(model.named_parameters())
        for k in full:
            store["step/grad_full/" + k] = prm[k].grad.detach().numpy().astype(np.float32)
        store["step/grad_full_names"] = np.array(full)
        store["step/param_names"] = np.array(names)
        store["step/grad_stats"] = np.array(gstat)     # per parameter: sum, abs-sum, first / middle / last element of the gradient
        store["step/update_stats"] = np.array(pstat)   # per parameter: sum after the step, abs-sum / first / middle / last of (after - before)
        bn_names, bn_stat = [], []
        for k, v in model.state_dict().items():
            if k.endswith("running_mean") or k.endswith("running_var"):
                bn_names.append(k)
                bn_stat.append([v.double().sum().item(), (v.double() - before[k].double()).abs().sum().item(), v.double().flatten()[0].item()])
        store["step/bn_names"] = np.array(bn_names)
        store["step/bn_stats"] = np.array(bn_stat)
        print("captured step statistics for %d parameters, %d BN buffers; dropout keeps %d of %d" % (len(names), len(bn_names), drop["mask"].sum(), drop["mask"].size))
    store[prefix + "loss_weight"] = np.array([loss_kwargs["loss_info"][h]["weight"] for h in heads], np.float64)


def main():
    store = {}
    # fp32 noise yardstick for the element-wise gradient comparison: the reference's OWN step through torch's two CPU convolution
    # back ends (oneDNN / native) -- same arithmetic, different summation order
    torch.backends.mkldnn.enabled = False
    alt = {}
    run_case("paramset/", None, alt)
    torch.backends.mkldnn.enabled = True
    run_case("paramset/", None, store)
    for k in [str(x) for x in store["step/grad_full_names"]]:
        a, b = store["step/grad_full/" + k].astype(np.float64), alt["step/grad_full/" + k].astype(np.float64)
        store["step/grad_full_noise/" + k] = np.float64(np.abs(a - b).max() / max(np.abs(a).max(), 1e-30))
    return_after = store
    run_case("typew1/", 1.0, store)
    run_case("wmap/", None, store, weight_maps=True)
    path = os.path.join(ROOT, "tests", "golden", "train_loss.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
