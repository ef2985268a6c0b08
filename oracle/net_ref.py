"""ORACLE (test infrastructure only) -- CPU restatement of the Cerberus network path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product (cerberus_amd/) never does.

Plain PyTorch-CPU fp32 functional restatement of
  * NetDesc.forward            reference models/net_desc.py:144-200
  * ResNet._forward_impl       reference models/backbone/resnet.py:273-286 (+ BasicBlock :81-97)
  * ConvBlock / _ConvLayer     reference models/utils/conv_layers.py:24-103
  * classification head        reference models/utils/net_layers.py:31-38
  * upsample2x                 reference models/utils/net_layers.py:45-46
  * infer_step                 reference models/run_desc.py:439-502
driven directly by a state dict (name -> array) with the reference's key names, so
no nn.Module of the reference is needed.  Pinned against the reference itself by
oracle/gen_golden_net.py (fixtures in tests/golden/net_*.npz); the reference ships
no tests for this path, so those fixtures are the only pin.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

RESNET34_LAYERS = (3, 4, 6, 3)
BN_EPS = 1e-5

HEAD_NAME_MAP = {  # models/run_desc.py:466-473
    "Gland": "Gland-INST",
    "Gland#TYPE": "Gland-TYPE",
    "Lumen": "Lumen-INST",
    "Nuclei": "Nuclei-INST",
    "Nuclei#TYPE": "Nuclei-TYPE",
    "Patch-Class": "Patch-Class",
}


def _t(sd, k):
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))


_TRAINING = False  # net_forward(training=True): BatchNorm on batch statistics (running buffers updated in place), autograd on
_EVAL_BN = ()      # net_forward(eval_bn_prefixes=...): BatchNorm layers under these state-dict prefixes stay in eval mode while training
                   # (the reference's _freeze_weight puts the frozen modules' BatchNorm2d in eval(), models/net_desc.py:105-121)


def _bn(sd, p, x):
    return F.batch_norm(
        x, _t(sd, p + ".running_mean"), _t(sd, p + ".running_var"), _t(sd, p + ".weight"), _t(sd, p + ".bias"),
        training=_TRAINING and not any(p.startswith(e) for e in _EVAL_BN), momentum=0.1, eps=BN_EPS,
    )


def cropping_center_nchw(x, crop_shape):
    """models/utils/misc_utils.py:6-25 (batch=True)."""
    h0 = int((x.shape[2] - crop_shape[0]) * 0.5)
    w0 = int((x.shape[3] - crop_shape[1]) * 0.5)
    return x[:, :, h0:h0 + crop_shape[0], w0:w0 + crop_shape[1]]


def backbone_forward(sd, x):
    """resnet.py:273-286. conv1 is 7x7 *stride 1* (resnet.py:195-197)."""
    x = F.conv2d(x, _t(sd, "backbone.conv1.weight"), None, stride=1, padding=3)
    x = F.relu(_bn(sd, "backbone.bn1", x))
    x0 = x
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    feats = [x0]
    for li, nblk in enumerate(RESNET34_LAYERS):
        for b in range(nblk):
            p = "backbone.layer%d.%d" % (li + 1, b)
            stride = 2 if (b == 0 and li > 0) else 1
            idt = x
            out = F.conv2d(x, _t(sd, p + ".conv1.weight"), None, stride=stride, padding=1)
            out = F.relu(_bn(sd, p + ".bn1", out))
            out = F.conv2d(out, _t(sd, p + ".conv2.weight"), None, stride=1, padding=1)
            out = _bn(sd, p + ".bn2", out)
            if (p + ".downsample.0.weight") in sd:
                idt = F.conv2d(x, _t(sd, p + ".downsample.0.weight"), None, stride=stride)
                idt = _bn(sd, p + ".downsample.1", idt)
            x = F.relu(out + idt)
        feats.append(x)
    return feats  # [x0, x1, x2, x3, x4]


def net_forward(sd, imgs_nchw, decoder_kwargs, considered_tasks, return_feats=False, training=False, eval_bn_prefixes=(), block_local_grads=(),
                dropout_scale=None):
    """NetDesc.forward (net_desc.py:144-200). imgs: float NCHW in 0..255.  training=True: model.train() semantics (batch-statistics
    BatchNorm, Dropout(0.3) in the Patch-Class branch, autograd enabled) -- the forward half of oracle/train_step_ref.py.
    eval_bn_prefixes: BatchNorm layers that stay in eval mode (frozen modules); block_local_grads: decoder names run the way the reference
    runs a decoder that is NOT in train_decoder_list -- under set_grad_enabled(False), with every conv layer switching autograd back on inside
    itself (models/utils/conv_layers.py:44-53), so gradients exist inside each block and stop at the skip + upsample sum (net_desc.py:182);
    dropout_scale: [N, 512] multiplier replacing the random Patch-Class dropout (keep / 0.7), for reproducible comparisons."""
    global _TRAINING, _EVAL_BN
    _TRAINING = bool(training)
    _EVAL_BN = tuple(eval_bn_prefixes)
    with torch.set_grad_enabled(bool(training)):
        imgs = imgs_nchw / 255.0
        feat_list = backbone_forward(sd, imgs)
        bottom = feat_list[-1]
        feat_list = list(feat_list)
        feat_list[-1] = F.conv2d(bottom, _t(sd, "conv_map.weight"), None)
        out = OrderedDict()
        for name, heads in decoder_kwargs.items():
            if name not in considered_tasks:
                continue
            if name == "Patch-Class":
                bf = bottom
                # net_desc.py:171-174 -- crops only when *both* dims != 9
                if bf.shape[2] != 9 and bf.shape[3] != 9:
                    bf = cropping_center_nchw(bf, [9, 9])
                v = F.adaptive_avg_pool2d(bf, (1, 1))
                p = "decoder_head.Patch-Class"
                v = F.relu(_bn(sd, p + ".bn1", v))
                if dropout_scale is not None:
                    v = v * torch.as_tensor(dropout_scale, dtype=v.dtype).reshape(v.shape[0], 512, 1, 1)
                else:
                    v = F.dropout(v, 0.3, training=_TRAINING)  # net_desc.py:70
                v = F.conv2d(v, _t(sd, p + ".conv1.weight"), _t(sd, p + ".conv1.bias"))
                v = F.relu(_bn(sd, p + ".bn2", v))
                v = F.conv2d(v, _t(sd, p + ".conv2.weight"), _t(sd, p + ".conv2.bias"))
                out[name] = v
                continue
            prev = feat_list[-1]
            for idx in range(1, 5):
                prev = F.interpolate(prev, scale_factor=2, mode="bilinear", align_corners=False)
                new = feat_list[-(idx + 1)] + prev
                if name in block_local_grads:
                    new = new.detach()
                for j in range(2):
                    p = "decoder_head.%s.%d.block.%d" % (name, idx - 1, j)
                    new = F.conv2d(new, _t(sd, p + ".conv.weight"), _t(sd, p + ".conv.bias"), padding=1)
                    new = F.relu(_bn(sd, p + ".bn", new))
                prev = new
            for clf in heads.keys():
                p = "output_head.%s.%s.x" % (name, clf)
                h = F.conv2d(prev, _t(sd, p + ".0.block.0.conv.weight"), _t(sd, p + ".0.block.0.conv.bias"))
                h = F.relu(_bn(sd, p + ".0.block.0.bn", h))
                h = F.conv2d(h, _t(sd, p + ".1.conv.weight"), _t(sd, p + ".1.conv.bias"))
                out[name.split("#")[0] + "-" + clf] = h
    _TRAINING = False
    _EVAL_BN = ()
    if return_feats:
        return out, feat_list, bottom
    return out


def infer_step(sd, img_list_nhwc_u8, output_shape, head_name_list, decoder_kwargs):
    """models/run_desc.py:439-502 on CPU; returns list of per-sample dicts of numpy arrays."""
    img = torch.as_tensor(img_list_nhwc_u8).type(torch.float32).permute(0, 3, 1, 2).contiguous()
    if not isinstance(output_shape, (list, tuple)):
        output_shape = [output_shape, output_shape]
    pred = net_forward(sd, img, decoder_kwargs, head_name_list)
    pred = OrderedDict((k, v.permute(0, 2, 3, 1).contiguous()) for k, v in pred.items())
    sub = OrderedDict()
    for hn_ in head_name_list:
        hn = HEAD_NAME_MAP[hn_]
        x = pred[hn]
        if hn == "Patch-Class":
            x = torch.argmax(torch.softmax(x, -1), dim=-1, keepdim=True)
            x = F.interpolate(x.type(torch.float32), size=list(output_shape), mode="nearest")
            x = torch.squeeze(x)
            if x.dim() == 2:
                x = x.unsqueeze(0)
        else:
            x = torch.softmax(x, -1)
            if "INST" in hn:
                x = x[..., 1:]
            h0 = int((x.shape[1] - output_shape[0]) * 0.5)  # misc/utils.py:94-104
            w0 = int((x.shape[2] - output_shape[1]) * 0.5)
            x = x[:, h0:h0 + output_shape[0], w0:w0 + output_shape[1]]
        if "TYPE" in hn:
            x = torch.argmax(x, dim=-1)
        sub[hn] = x.numpy()
    return [
        {k: v[i] for k, v in sub.items()} for i in range(img.shape[0])
    ]
