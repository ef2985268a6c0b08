import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs, make_state_dict
m = create_model(**default_model_kwargs()); m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}, strict=True)
for win, osz, n in ((256, 256, 3), (448, 144, 2), (272, 272, 2), (208, 80, 2)):
    tiles = torch.from_numpy(np.random.RandomState(win).randint(0, 256, (n, win, win, 3)).astype(np.uint8)).cuda()
    m.set_planar(1); ref = {k: v.clone() for k, v in m.infer_tiles(tiles, osz).items()}
    m.set_planar(2); got = {k: v.clone() for k, v in m.infer_tiles(tiles, osz).items()}
    torch.cuda.synchronize()
    for k in ref:
        if ref[k].dtype.is_floating_point: print(win, osz, k, "max abs diff %.3e" % (got[k] - ref[k]).abs().max().item())
        else: print(win, osz, k, "mismatch frac %.2e" % (got[k] != ref[k]).float().mean().item())
