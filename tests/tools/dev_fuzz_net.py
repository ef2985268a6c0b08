"""Developer fuzz: random tile geometries / batch sizes / crops, HIP forward + output wrapper vs the torch-CPU oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cerberus_amd.net_desc import create_model
from cerberus_amd.run_desc import infer_step
from cerberus_amd.weights import default_model_kwargs, make_state_dict
from oracle import net_ref

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
worst = 0.0
for i in range(n_cases):
    tasks = [None, ["Nuclei"], ["Gland", "Lumen"]][rs.randint(3)]
    kw = default_model_kwargs(tasks)
    sd = {k: torch.from_numpy(v) for k, v in make_state_dict(int(rs.randint(5)), kw["decoder_kwargs"], kw["considered_tasks"]).items()}
    m = create_model(**kw); m.load_state_dict(sd, strict=True)
    h, w = 16 * int(rs.randint(1, 26)), 16 * int(rs.randint(1, 26))
    n = int(rs.randint(1, 4))
    out = [int(rs.randint(2, h + 1)), int(rs.randint(2, w + 1))] if rs.randint(2) else [h, w]  # a side of 1 trips the reference's own squeeze (run_desc.py:484-487)
    algo = [0, 1, 5, 6, 7, 6, 7, 5][int(rs.randint(8))]  # direct, F(2x2), and the three F(4x4) choices
    m.set_conv_algo(algo)
    tiles = rs.randint(0, 256, (n, h, w, 3)).astype(np.uint8)
    got = infer_step(torch.from_numpy(tiles), m, out, kw["considered_tasks"])
    ref = net_ref.infer_step(sd, tiles, out, kw["considered_tasks"], kw["decoder_kwargs"])
    err, mis = 0.0, 0.0
    for a, b in zip(got, ref):
        for k in b:
            assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype, (k, a[k].shape, b[k].shape)
            if b[k].dtype == np.float32:
                err = max(err, float(np.abs(a[k] - b[k]).max()))
            else:
                mis = max(mis, float((a[k] != b[k]).mean()))
    worst = max(worst, err)
    print("case %2d: tasks %-18s n %d tile %3dx%3d out %s algo %d  max prob err %.2e  max int mismatch frac %.1e %s" % (
        i, "all" if tasks is None else ",".join(tasks), n, h, w, out, algo, err, mis, "FAIL" if err > 1e-4 or mis > 1e-3 else ""), flush=True)
print("worst prob err %.2e" % worst)
