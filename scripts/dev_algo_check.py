"""debug: max |difference| of the probability maps between conv algorithms (first argument vs the rest), batch of 256-pixel tiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs, make_state_dict
m = create_model(**default_model_kwargs())
m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}, strict=True)
m._ensure_handle()
algos = [int(a) for a in sys.argv[1:]] or [1, 7]
for win, osz, n in ((256, 256, 8), (448, 144, 3), (96, 96, 5), (144, 144, 3), (176, 80, 5), (208, 208, 2), (304, 144, 3)):
    tiles = torch.from_numpy(np.random.RandomState(3).randint(0, 256, (n, win, win, 3)).astype(np.uint8)).cuda()
    ref = None
    for algo in algos:
        m.set_conv_algo(algo)
        a = {k: v.clone() for k, v in m.infer_tiles(tiles, osz).items()}
        b = {k: v.clone() for k, v in m.infer_tiles(tiles, osz).items()}
        rep = max((a[k].float() - b[k].float()).abs().max().item() for k in a)
        if ref is None:
            ref = a
        else:
            print(win, osz, "algo", algo, "vs", algos[0], {k: float("%.3g" % (a[k].float() - ref[k].float()).abs().max().item()) for k in a if a[k].dtype.is_floating_point}, "repeat", rep, flush=True)
