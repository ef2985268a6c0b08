"""State-dict schema of the Cerberus network and a seeded synthetic-weight recipe.

The schema mirrors what ``NetDesc.state_dict()`` of the reference produces
(reference: models/net_desc.py:23-103, models/backbone/resnet.py:195-271,
models/utils/net_layers.py:10-38, models/utils/conv_layers.py:24-103): 558 keys
for the shipped six-head configuration.  No checkpoint can travel to the GPU box
(no network), so tests / bench use weights drawn from ``numpy.random.RandomState``
with the recipe below.  The recipe is *non-saturating*: with the reference's own
``weights_init_cnn`` the softmax outputs are all exactly 0 or 1 (SURVEY.md par.7
step 0), which would make a 1e-4 probability tolerance vacuous.

Only numpy is used so the same bytes are produced in the py3.10/torch container,
the conda py3.9 interpreter and on the GPU box.
"""
from collections import OrderedDict

import numpy as np

RESNET34_LAYERS = (3, 4, 6, 3)
FILTERS = (64, 64, 128, 256, 512)  # models/backbone/__init__.py:31

DEFAULT_DECODER_KWARGS = OrderedDict(
    [
        ("Lumen", OrderedDict(INST=3)),
        ("Gland", OrderedDict(INST=3)),
        ("Nuclei", OrderedDict(INST=3)),
        ("Nuclei#TYPE", OrderedDict(TYPE=7)),
        ("Gland#TYPE", OrderedDict(TYPE=3)),
        ("Patch-Class", OrderedDict(OUT=9)),
    ]
)  # models/paramset.yml:46-60

DEFAULT_REQ_TARGET_CODE = OrderedDict(
    [
        ("Lumen-INST", "IP-ERODED-CONTOUR-3"),
        ("Gland-INST", "IP-ERODED-CONTOUR-11"),
        ("Nuclei-INST", "IP-ERODED-CONTOUR-3"),
        ("Nuclei-TYPE", "TP"),
        ("Gland-TYPE", "TP"),
        ("Patch-Class", "PC"),
    ]
)  # models/paramset.yml:37-43


def default_model_kwargs(considered_tasks=None):
    kw = {
        "encoder_backbone_name": "resnet34",
        "decoder_kwargs": OrderedDict((k, OrderedDict(v)) for k, v in DEFAULT_DECODER_KWARGS.items()),
    }
    kw["considered_tasks"] = (
        list(DEFAULT_DECODER_KWARGS.keys()) if considered_tasks is None else list(considered_tasks)
    )
    return kw


def _bn(prefix, ch):
    return [
        (prefix + ".weight", (ch,), "bn_w"),
        (prefix + ".bias", (ch,), "bn_b"),
        (prefix + ".running_mean", (ch,), "bn_m"),
        (prefix + ".running_var", (ch,), "bn_v"),
        (prefix + ".num_batches_tracked", (), "bn_n"),
    ]


def state_dict_schema(decoder_kwargs=None, considered_tasks=None):
    """List of (key, shape, kind) in the reference's registration order."""
    decoder_kwargs = DEFAULT_DECODER_KWARGS if decoder_kwargs is None else decoder_kwargs
    if considered_tasks is None:
        considered_tasks = list(decoder_kwargs.keys())
    out = [("backbone.conv1.weight", (64, 3, 7, 7), "conv")]
    out += _bn("backbone.bn1", 64)
    inpl = 64
    for li, nblk in enumerate(RESNET34_LAYERS):
        planes = FILTERS[li + 1]
        for b in range(nblk):
            p = "backbone.layer%d.%d" % (li + 1, b)
            stride = 2 if (b == 0 and li > 0) else 1
            out.append((p + ".conv1.weight", (planes, inpl, 3, 3), "conv"))
            out += _bn(p + ".bn1", planes)
            out.append((p + ".conv2.weight", (planes, planes, 3, 3), "conv"))
            out += [(k, sh, "bn_w_res" if kd == "bn_w" else kd) for (k, sh, kd) in _bn(p + ".bn2", planes)]
            if stride != 1 or inpl != planes:
                out.append((p + ".downsample.0.weight", (planes, inpl, 1, 1), "conv"))
                out += _bn(p + ".downsample.1", planes)
            inpl = planes
    out.append(("backbone.fc.weight", (1000, 512), "fc"))
    out.append(("backbone.fc.bias", (1000,), "bias"))
    out.append(("conv_map.weight", (256, 512, 1, 1), "conv"))

    f = FILTERS
    dec_ch = [(f[-2], (f[-2], f[-3])), (f[-3], (f[-3], f[-4])), (f[-4], (f[-4], f[-5])), (f[-5], (f[-5], f[-5]))]
    dec_entries, head_entries = [], []
    for name, heads in decoder_kwargs.items():
        if name not in considered_tasks:
            continue
        if name == "Patch-Class":
            out_ch = list(heads.values())[-1]
            p = "decoder_head.Patch-Class"
            dec_entries += _bn(p + ".bn1", 512)
            dec_entries += [(p + ".conv1.weight", (256, 512, 1, 1), "conv"), (p + ".conv1.bias", (256,), "bias")]
            dec_entries += _bn(p + ".bn2", 256)
            dec_entries += [(p + ".conv2.weight", (out_ch, 256, 1, 1), "conv_out"), (p + ".conv2.bias", (out_ch,), "bias")]
            continue
        for u, (cin, units) in enumerate(dec_ch):
            c = cin
            for j, cout in enumerate(units):
                p = "decoder_head.%s.%d.block.%d" % (name, u, j)
                dec_entries += _bn(p + ".bn", cout)
                dec_entries += [(p + ".conv.weight", (cout, c, 3, 3), "conv"), (p + ".conv.bias", (cout,), "bias")]
                c = cout
        for clf, out_ch in heads.items():
            p = "output_head.%s.%s.x" % (name, clf)
            head_entries += _bn(p + ".0.block.0.bn", 96)
            head_entries += [(p + ".0.block.0.conv.weight", (96, 64, 1, 1), "conv"), (p + ".0.block.0.conv.bias", (96,), "bias")]
            head_entries += [(p + ".1.conv.weight", (out_ch, 96, 1, 1), "conv_out"), (p + ".1.conv.bias", (out_ch,), "bias")]
    return out + dec_entries + head_entries


def make_state_dict(seed=0, decoder_kwargs=None, considered_tasks=None, head_logit_scale=None):
    """Seeded, non-saturating fp32 weights as an OrderedDict[str, np.ndarray].

    conv: N(0, 2/fan_in) (keeps ReLU activations O(1) through ~40 layers);
    final 1x1 of every head: N(0, 6/fan_in) so the logits spread enough for
    argmax / 0.5-threshold decisions to be non-trivial but unsaturated;
    BN: gamma U(0.7,1.1) (U(0.2,0.4) for the second BN of every residual block),
    beta N(0,0.1), running_mean N(0,0.1), running_var U(0.8,1.25).

    head_logit_scale: optional {"<decoder>.<head>": float32 factor} -- the LAST 1x1 convolution (weight and bias) of that output head is
    multiplied by the factor in float32, i.e. its logits are exactly `factor` times the unscaled recipe's up to rounding: the "confident
    model" families of the golden-vector generator (calibration logits moved to ~30 / ~80 with everything upstream unchanged).
    """
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    for key, shape, kind in state_dict_schema(decoder_kwargs, considered_tasks):
        if kind in ("conv", "conv_out"):
            fan_in = int(np.prod(shape[1:]))
            gain = 2.0 if kind == "conv" else 6.0
            w = rs.standard_normal(shape) * np.sqrt(gain / fan_in)
        elif kind == "fc":
            w = rs.standard_normal(shape) * 0.01
        elif kind == "bias":
            w = rs.standard_normal(shape) * 0.1
        elif kind == "bn_w":
            w = rs.uniform(0.7, 1.1, shape)
        elif kind == "bn_w_res":  # residual-branch BN: keeps the 16-block trunk from doubling its variance per block
            w = rs.uniform(0.2, 0.4, shape)
        elif kind == "bn_b":
            w = rs.standard_normal(shape) * 0.1
        elif kind == "bn_m":
            w = rs.standard_normal(shape) * 0.1
        elif kind == "bn_v":
            w = rs.uniform(0.8, 1.25, shape)
        elif kind == "bn_n":
            sd[key] = np.array(1, dtype=np.int64)
            continue
        else:  # pragma: no cover
            raise ValueError(kind)
        sd[key] = np.ascontiguousarray(w, dtype=np.float32)
    for name, factor in (head_logit_scale or {}).items():
        dec, clf = name.rsplit(".", 1)
        for leaf in ("weight", "bias"):
            key = "output_head.%s.%s.x.1.conv.%s" % (dec, clf, leaf)
            sd[key] = np.ascontiguousarray(sd[key] * np.float32(factor), dtype=np.float32)
    return sd


def reference_init_state_dict(decoder_kwargs=None, considered_tasks=None, generator=None):
    """What the reference's NetDesc.__init__ leaves in a freshly built model (models/net_desc.py:89-103 `weights_init_cnn`,
    models/backbone/resnet.py:215-220): conv weights kaiming-normal (fan_out, relu), conv / linear biases at torch's defaults,
    BatchNorm weight 1 / bias 0 / running_mean 0 / running_var 1 / num_batches_tracked 0.  Drawn from torch's global generator (or
    `generator`), so `torch.manual_seed` governs it as it governs the reference; the streams are not bit-identical (module order)."""
    import math

    import torch

    sd = OrderedDict()
    fan_in_of = {}
    for key, shape, kind in state_dict_schema(decoder_kwargs, considered_tasks):
        if kind in ("conv", "conv_out"):
            fan_out = int(shape[0] * np.prod(shape[2:]))
            fan_in_of[key.rsplit(".", 1)[0]] = int(np.prod(shape[1:]))
            w = torch.randn(tuple(shape), generator=generator) * math.sqrt(2.0 / fan_out)
        elif kind == "fc":  # backbone.fc (never called): nn.Linear default
            fan_in_of[key.rsplit(".", 1)[0]] = int(shape[1])
            b = 1.0 / math.sqrt(shape[1])
            w = (torch.rand(tuple(shape), generator=generator) * 2 - 1) * b
        elif kind == "bias":
            if key.startswith("backbone.fc"):
                w = torch.zeros(tuple(shape))
            else:
                b = 1.0 / math.sqrt(max(1, fan_in_of.get(key.rsplit(".", 1)[0], 1)))
                w = (torch.rand(tuple(shape), generator=generator) * 2 - 1) * b
        elif kind in ("bn_w", "bn_w_res", "bn_v"):
            w = torch.ones(tuple(shape))
        elif kind in ("bn_b", "bn_m"):
            w = torch.zeros(tuple(shape))
        elif kind == "bn_n":
            sd[key] = np.array(0, dtype=np.int64)
            continue
        else:  # pragma: no cover
            raise ValueError(kind)
        sd[key] = np.ascontiguousarray(w.numpy(), dtype=np.float32)
    return sd


def state_dict_sha256(sd):
    import hashlib

    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v).tobytes())
    return h.hexdigest()
