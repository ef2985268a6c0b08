"""Pin oracle/net_ref.py harder: random tile geometries / batch sizes / output crops / task subsets through the REFERENCE's own
NetDesc + infer_step (CPU) and through the oracle restatement.  TEST INFRASTRUCTURE; runs only in the build container.

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/fuzz_ref_vs_oracle_net.py [n_cases] [seed]"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden_net as G  # noqa: E402  (sets up the stubs and imports the reference; its __main__ block does not run)
import numpy as np  # noqa: E402
import torch  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
worst = 0.0
for i in range(n_cases):
    tasks = [None, ["Nuclei"], ["Gland", "Lumen"], ["Gland", "Lumen", "Nuclei"], ["Patch-Class", "Nuclei"]][rs.randint(5)]
    kw = G.default_model_kwargs(tasks)
    sd_np = G.make_state_dict(int(rs.randint(4)), kw["decoder_kwargs"], kw["considered_tasks"])
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    model = G.create_model(**kw)
    model.load_state_dict(sd, strict=True)
    model.eval()
    h, w = 16 * int(rs.randint(1, 14)), 16 * int(rs.randint(1, 14))
    n = int(rs.randint(1, 4))
    out = int(rs.randint(1, min(h, w) + 1)) if rs.randint(2) else [int(rs.randint(1, h + 1)), int(rs.randint(1, w + 1))]
    tiles = rs.randint(0, 256, (n, h, w, 3)).astype(np.uint8)
    ref = G.ref_infer_step(torch.from_numpy(tiles), model, out, kw["considered_tasks"])
    orc = G.net_ref.infer_step(sd, tiles, out, kw["considered_tasks"], kw["decoder_kwargs"])
    err = 0.0
    assert len(ref) == len(orc) == n
    for a, b in zip(ref, orc):
        assert list(a.keys()) == list(b.keys()), (list(a.keys()), list(b.keys()))
        for k in a:
            assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype, (k, a[k].shape, b[k].shape, a[k].dtype, b[k].dtype)
            if a[k].dtype == np.float32:
                err = max(err, float(np.abs(a[k] - b[k]).max()))
            else:
                assert (a[k] != b[k]).mean() < 1e-3, k
    worst = max(worst, err)
    print("case %2d: tasks %-22s n %d tile %3dx%3d out %-10s max |ref - oracle| %.1e" % (i, "all" if tasks is None else ",".join(tasks), n, h, w, out, err), flush=True)
    assert err < 1e-5
print("reference vs oracle (network + infer_step): %d cases, worst %.1e" % (n_cases, worst))
