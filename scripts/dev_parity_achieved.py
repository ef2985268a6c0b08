"""Developer tool (GPU): what the default path ACHIEVES on every reference-generated network fixture -- max |got - ref_fp32| and
max |got - ref_fp64| per family and INST head -- as JSON for the ACHIEVED table of tests/test_net_gpu.py and DESIGN par.5."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_net_gpu as T
from cerberus_amd.run_desc import infer_step
res = {}
for tag in ["cfg1_nuclei", "cfg2_all", "g448_all", "small96_all", "seed1_all", "refinit_all"]:
    g = np.load(os.path.join(ROOT, "tests", "golden", "net_%s.npz" % tag))
    m, sd, kw, tasks = T._golden_model(g)
    n, hw, osz = int(g["n"]), int(g["hw"]), int(g["out_shape"])
    tiles = np.random.RandomState(int(g["tile_seed"])).randint(0, 256, (n, hw, hw, 3)).astype(np.uint8)
    for algo in (6, 5, 7, 1, 0):
        m.set_conv_algo(algo)
        out = infer_step(torch.from_numpy(tiles), m, osz, tasks)
        for k in out[0]:
            a = np.stack([out[i][k] for i in range(n)])
            if a.dtype != np.float32 or not k.endswith("INST"):
                continue
            key = "out_crops/" + k
            ref = g[key] if key in g else g["out_full/" + k]
            got = T._crops(a) if key in g else a
            p64 = g["p64_crops/" + k] if ("p64_crops/" + k) in g else g["p64_full/" + k]
            res.setdefault(tag, {}).setdefault(k, {})["algo%d" % algo] = [float(np.abs(got - ref).max()), float(np.abs(got - p64).max()), float(np.abs(ref - p64).max())]
    m.set_conv_algo(6)
print(json.dumps(res, indent=1))
