"""A checkpoint directory in the reference's layout (settings.yml + weights.tar, run_infer_wsi.py --model) holding the package's seeded test weights
with every INST head's background bias raised so that ~q of the given tiles' pixels are foreground (bench.py's sparse_foreground_weights, calibrated on
the caller's own texture): the plain seeded weights paint slide-sized blobs, which no labelling window can hold, so command-line runs on them can only
compare class maps.  Needs the GPU (one forward for the calibration)."""
import json
import os

import numpy as np
import torch


def stain_atlas(n=64, tile=256, seed=17):
    """n stain-field tiles with +-10 of noise -- the texture of the synthetic TIFF slides (bench.py's ingest leg, scripts/dev_r06_giant_slide.py)."""
    from cerberus_amd.synth_tiles import stain_field

    rs = np.random.RandomState(seed)
    return [np.clip(stain_field(tile, 100 + i).astype(np.int16) + rs.randint(-10, 11, (tile, tile, 3)), 0, 255).astype(np.uint8) for i in range(n)]


def write_sparse_model_dir(path, tiles, q=0.02):
    """tiles: uint8 [n, 256, 256, 3]; q: foreground share, a float or {decoder name: share, "default": share}.  -> {head key: bias shift}"""
    import yaml

    from cerberus_amd.net_desc import create_model
    from cerberus_amd.weights import DEFAULT_REQ_TARGET_CODE, default_model_kwargs, make_state_dict

    kw = default_model_kwargs()
    sd = {k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}
    m = create_model(**kw)
    m.load_state_dict(sd, strict=True)
    lg = m(torch.from_numpy(np.ascontiguousarray(tiles)).cuda())
    shifts = {}
    for name, hname, och, key in m._decoders:
        if hname != "INST":
            continue
        v = lg[key]  # (n, 3, H, W)
        margin = v[:, 1] - torch.logsumexp(torch.stack([v[:, 0], v[:, 2]]), 0)
        flat = margin.flatten().float()
        qq = q.get(name, q.get("default", 0.02)) if isinstance(q, dict) else q
        d = float(torch.quantile(flat[:: max(1, flat.numel() // 1000000)], 1.0 - qq))
        sd["output_head.%s.INST.x.1.conv.bias" % name][0] += d
        shifts[key] = round(d, 4)
    os.makedirs(path, exist_ok=True)
    torch.save({"desc": sd}, os.path.join(path, "weights.tar"))
    plain = json.loads(json.dumps({"dataset_kwargs": {"req_target_code": DEFAULT_REQ_TARGET_CODE}, "model_kwargs": kw}))
    with open(os.path.join(path, "settings.yml"), "w") as fh:
        yaml.safe_dump(plain, fh, sort_keys=False)
    del m, lg
    torch.cuda.empty_cache()
    return shifts
