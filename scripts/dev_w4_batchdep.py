"""debug: does conv_algo 5 give the same tile values under different batch compositions / repeated runs (448 -> 144 crops)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs, make_state_dict
m = create_model(**default_model_kwargs())
m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}, strict=True)
m._ensure_handle()
win, osz = int(sys.argv[1]) if len(sys.argv) > 1 else 448, int(sys.argv[2]) if len(sys.argv) > 2 else 144
tiles = torch.from_numpy(np.random.RandomState(3).randint(0, 256, (7, win, win, 3)).astype(np.uint8)).cuda()
for algo in (1, 5):
    m.set_conv_algo(algo)
    a = {k: v.clone() for k, v in m.infer_tiles(tiles, osz).items()}
    b = {k: v.clone() for k, v in m.infer_tiles(tiles, osz).items()}
    c = {k: v.clone() for k, v in m.infer_tiles(tiles[:6], osz).items()}
    d = {k: v.clone() for k, v in m.infer_tiles(tiles[1:2], osz).items()}
    for k in a:
        if a[k].dtype.is_floating_point:
            print(algo, k, "repeat", (a[k] - b[k]).abs().max().item(), "n7 vs n6", (a[k][:6] - c[k]).abs().max().item(), "n7 vs single", (a[k][1:2] - d[k]).abs().max().item(),
                  "where", torch.nonzero((a[k][:6] != c[k]).flatten(1).any(1)).flatten().tolist())
