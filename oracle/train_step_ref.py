"""ORACLE (test infrastructure; bench.py's cpu_baseline leg for --mode train): the reference's whole training step on the CPU as a
torch-autograd restatement -- train-mode forward (oracle/net_ref.py, batch-statistics BatchNorm + Dropout), the six head losses
(oracle/train_ref.py, models/run_desc.py:88-170), backward, Adam (models/opt.py:47-58).  Reported, not optimised."""
from collections import OrderedDict

import numpy as np
import torch

from . import net_ref, train_ref


def make_step(sd, decoder_kwargs, considered_tasks, lr=1.0e-4):
    """sd: state dict (numpy / tensors, reference key names).  -> step(imgs_u8_nhwc, targets {head: [N,H,W,1] or [N,1,1,1]}, has [N, heads] bool)"""
    params = OrderedDict()
    for k, v in sd.items():
        t = torch.as_tensor(np.asarray(v)).clone()
        if t.dtype == torch.float32 and not k.endswith(("running_mean", "running_var")) and not k.startswith("backbone.fc"):
            t.requires_grad_(True)
        params[k] = t
    opt = torch.optim.Adam([p for p in params.values() if p.requires_grad], lr=lr, betas=(0.9, 0.999))

    def step(imgs_u8, targets, has):
        opt.zero_grad()
        x = torch.as_tensor(imgs_u8).float().permute(0, 3, 1, 2).contiguous()
        logits = net_ref.net_forward(params, x, decoder_kwargs, considered_tasks, training=True)
        total = 0
        for j, (h, lg) in enumerate(logits.items()):
            total = total + train_ref.head_loss_tensor(h, lg, targets[h], has[:, j], n_classes=lg.shape[1])
        total.backward()
        opt.step()
        return float(total.item())

    return step
