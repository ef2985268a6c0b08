"""A/B of the 3x3 convolution algorithms (cerb_net_set_conv_algo 1 = conv_wino F(2x2), 5 / 7 / 6 = F(4x4)) inside the configs[1] batch step (GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs, make_state_dict

dev = torch.device("cuda", 0)
m = create_model(**default_model_kwargs())
m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}, strict=True)
for algo in [int(a) for a in (sys.argv[1:] or ["1", "3", "4", "1", "3", "4"])]:
    m._ensure_handle()
    m.set_conv_algo(algo)
    dt, step, n = bench.batch_loop(m, dev, 0, 30, 5, None, "nccl")
    _, rows = bench.kernel_table(m, step, n)
    ws = [r for r in rows if r["kernel"].startswith("conv_wino")]
    print("conv_algo %d: step %.3f ms | %s" % (algo, dt / 30 * 1e3, ", ".join("%s x%d %.3f ms (exec %.1f TF, frac %.3f)" % (r["kernel"], r["launches"], r["ms_per_step"], r["achieved"], r["frac"]) for r in ws)), flush=True)
