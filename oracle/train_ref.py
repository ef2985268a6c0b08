"""ORACLE (test infrastructure): CPU restatement of the per-head loss of the reference's train_step (models/run_desc.py:88-170) and of
its loss functions (models/utils/loss_utils.py:6-21 xentropy_loss, :60-75 dice_loss), torch-CPU fp32 with autograd for the gradient.
Pinned by tests/golden/train_loss.npz, which oracle/gen_golden_train_loss.py produced by running the reference's own train_step.

Kept quirks (all in the reference): the TYPE heads' pixel weights are the class-weight map with 0 for background, so their cross
entropy only counts pixels inside objects (:118-124); their Dice term (positive classes, masked by target > 0, smooth 1e-3, summed
over classes, models/utils/loss_utils.py:60-75) is NOT multiplied by the per-sample target flags (:137-146); for Patch-Class the
weight map is taken before the squeeze, so `sample_loss * sample_wmap[:, 0]` broadcasts [N] against [N, 1, 1] and every sample's
loss becomes the mean over ALL samples (:126-128,152-154)."""
import numpy as np
import torch
import torch.nn.functional as F

PARAMSET_LOSS = {  # models/paramset.yml:13-31
    "loss_info": {"Lumen-INST": {"weight": 1.5, "loss": {"ce": 1}}, "Gland-INST": {"weight": 1.4, "loss": {"ce": 1}},
                  "Nuclei-INST": {"weight": 1, "loss": {"ce": 1}}, "Nuclei-TYPE": {"weight": 0, "loss": {"ce": 1, "dice": 1}},
                  "Gland-TYPE": {"weight": 1, "loss": {"ce": 1, "dice": 1}}, "Patch-Class": {"weight": 0.4, "loss": {"ce": 1}}},
    "class_weight": {"Gland-TYPE": {1: 1, 2: 1}, "Nuclei-TYPE": {1: 12, 2: 1, 3: 2, 4: 6, 5: 12, 6: 2}},
}


def head_loss(head_name, logits_nchw, target_nhw1, has_target, loss_opts=PARAMSET_LOSS, n_classes=None, weight_map_nhw1=None):
    """weight_map_nhw1: the head's '#WEIGHT-MAP' target (models/run_desc.py:111-117) or None.
    -> (loss value as train_step reports it = weighted head loss, d(that)/d(logits) as float32 NCHW numpy)"""
    pred = torch.tensor(np.asarray(logits_nchw), dtype=torch.float32, requires_grad=True)
    total = head_loss_tensor(head_name, pred, target_nhw1, has_target, loss_opts, n_classes, weight_map_nhw1)
    total.backward()
    return float(total.item()), pred.grad.numpy()


def head_loss_tensor(head_name, pred, target_nhw1, has_target, loss_opts=PARAMSET_LOSS, n_classes=None, weight_map_nhw1=None):
    """The weighted head loss as a tensor attached to `pred` (logits NCHW, possibly the output of a network under autograd)."""
    true = torch.tensor(np.asarray(target_nhw1), dtype=torch.float32).permute(0, 3, 1, 2).contiguous()  # NCHW like :60-62
    flag = torch.tensor(np.asarray(has_target).astype(np.float32))
    wmap = torch.ones_like(true)
    if weight_map_nhw1 is not None:
        wmap = torch.tensor(np.asarray(weight_map_nhw1), dtype=torch.float32).permute(0, 3, 1, 2).contiguous()
    binary = None
    if head_name in ("Nuclei-TYPE", "Gland-TYPE"):
        binary = (true > 0).float()
        wmap = true.clone()
        for cv, cw in loss_opts["class_weight"][head_name].items():
            wmap[true == cv] = cw
    p, t = pred, true
    if head_name == "Patch-Class":
        t, p = torch.squeeze(t), torch.squeeze(p)
    total = 0
    for name, w in loss_opts["loss_info"][head_name]["loss"].items():
        if name == "dice":
            nc = n_classes if n_classes is not None else pred.shape[1]
            oh = F.one_hot(torch.squeeze(t.to(torch.int64)), num_classes=nc).permute(0, 3, 1, 2)[:, 1:].float()
            sm = torch.softmax(p, 1)[:, 1:]
            inse = torch.sum(sm * oh * binary, (0, 2, 3))
            lsum = torch.sum(sm * binary, (0, 2, 3))
            rsum = torch.sum(oh * binary, (0, 2, 3))
            term = torch.sum(1.0 - (2.0 * inse + 1e-3) / (lsum + rsum + 1e-3))
        else:
            ce = F.cross_entropy(p, torch.squeeze(t).to(torch.int64), reduction="none")
            ce = ce * wmap[:, 0]
            ce = torch.mean(ce, dim=(1, 2))
            term = torch.sum(ce * flag) / (torch.sum(flag) + 1.0e-8)
        total = total + term * w
    return total * loss_opts["loss_info"][head_name]["weight"]
