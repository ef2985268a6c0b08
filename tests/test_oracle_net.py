"""Pins the CPU oracle of the network path (oracle/net_ref.py) to the golden vectors captured from the
reference itself (oracle/gen_golden_net.py).  Tolerances are loose only for cross-machine oneDNN rounding."""
import os

import numpy as np
import pytest
import torch

from cerberus_amd.weights import default_model_kwargs, make_state_dict
from oracle import net_ref

CROPS = [(0, 0), (96, 96), (192, 192)]
CS = 64


def _crops(a):
    return np.stack([a[:, y:y + CS, x:x + CS] for (y, x) in CROPS], axis=1)


@pytest.mark.parametrize("tag", ["cfg1_nuclei", "cfg2_all", "g448_all", "small96_all"])
def test_oracle_matches_reference_fixtures(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "net_%s.npz" % tag))
    tasks = [str(t) for t in g["tasks"]]
    kw = default_model_kwargs(tasks)
    sd = {k: torch.from_numpy(v) for k, v in make_state_dict(int(g["weight_seed"]), kw["decoder_kwargs"], tasks).items()}
    n, hw, osz = int(g["n"]), int(g["hw"]), int(g["out_shape"])
    tiles = np.random.RandomState(int(g["tile_seed"])).randint(0, 256, (n, hw, hw, 3)).astype(np.uint8)
    x = torch.from_numpy(tiles).float().permute(0, 3, 1, 2).contiguous()
    logits, feats, bottom = net_ref.net_forward(sd, x, kw["decoder_kwargs"], tasks, return_feats=True)
    for i, f in enumerate(feats[:4] + [bottom]):
        assert abs(f.double().mean().item() - float(g["feat_mean/x%d" % i])) < 1e-5
    for k, v in logits.items():
        a = v.permute(0, 2, 3, 1).contiguous().numpy()
        key = "logits_crops/" + k
        ref = g[key] if key in g else g["logits_full/" + k]
        got = _crops(a) if key in g else a
        assert np.abs(got - ref).max() < 2e-4, k
        assert abs(a.astype(np.float64).mean() - float(g["logits_mean/" + k])) < 1e-5
    out = net_ref.infer_step(sd, tiles, osz, tasks, kw["decoder_kwargs"])
    for k in out[0].keys():
        a = np.stack([out[i][k] for i in range(n)])
        assert str(a.dtype) == str(g["out_dtype/" + k])
        a4 = a[..., None] if a.ndim == 3 else a
        key = "out_crops/" + k
        ref = g[key] if key in g else g["out_full/" + k]
        got = _crops(a4) if key in g else a4
        if a.dtype == np.float32:
            assert np.abs(got - ref).max() < 1e-5, k
        else:
            assert (got != ref).mean() < 1e-3, k
