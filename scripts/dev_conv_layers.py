"""Per-layer times of the 3x3 convolutions under two algorithms, inside the configs[1] batch step (GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs, make_state_dict

dev = torch.device("cuda", 0)
m = create_model(**default_model_kwargs())
m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}, strict=True)
res = {}
for algo in [int(a) for a in (sys.argv[1:] or ["1", "5"])]:
    m._ensure_handle()
    m.set_conv_algo(algo)
    dt, step, n = bench.batch_loop(m, dev, 0, 10, 3, None, "nccl")
    acc = {}
    for rep in range(3):
        m.profile(True)
        step()
        torch.cuda.synchronize()
        for name, kern, fl, ms in m.profile_records():
            if kern.startswith("conv_wino"):
                a = acc.setdefault(name, [fl, 0.0, kern])
                a[1] += ms / 3
        m.profile(False)
    res[algo] = acc
algos = list(res)
print("%-44s %8s " % ("layer", "GFLOP") + " ".join("a%d ms  algTF" % a for a in algos))
for name in res[algos[0]]:
    fl = res[algos[0]][name][0]
    print("%-44s %8.1f " % (name, fl / 1e9) + " ".join("%6.3f %6.1f" % (res[a][name][1], fl / res[a][name][1] / 1e9) for a in algos))
