"""A/B of the output-head launch schemes (cerb_net_set_head_algo) inside the configs[1] batch step (GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs, make_state_dict

dev = torch.device("cuda", 0)
m = create_model(**default_model_kwargs())
m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}, strict=True)
for algo in [int(a) for a in (sys.argv[1:] or ["2", "1", "2", "1"])]:
    m._ensure_handle()
    m.set_head_algo(algo)
    dt, step, n = bench.batch_loop(m, dev, 0, 30, 5, None, "nccl")
    _, rows = bench.kernel_table(m, step, n)
    heads = [r for r in rows if r["kernel"].startswith("head")]
    print("head_algo %d: step %.3f ms | %s" % (algo, dt / 30 * 1e3, ", ".join("%s x%d %.3f ms (%.1f TF, frac %.3f)" % (r["kernel"], r["launches"], r["ms_per_step"], r["achieved"], r["frac"]) for r in heads)), flush=True)
