"""Developer check of conv_wino4s.hip (cerb_net_set_planar(2)): bitwise equality with conv_wino4p.hip (planar 1) and NHWC (planar 0), then A/B timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs, make_state_dict

dev = torch.device("cuda", 0)
m = create_model(**default_model_kwargs())
m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}, strict=True)
m._ensure_handle()
ok = True
quick = "--quick" in sys.argv
for (n, hw, osz) in (((3, 256, 256), (2, 448, 144)) if quick else ((3, 256, 256), (2, 448, 144), (5, 272, 272), (1, 304, [144, 160]))):
    tiles = torch.from_numpy(np.random.RandomState(hw).randint(0, 256, (n, hw, hw, 3)).astype(np.uint8)).cuda()
    res = {}
    for pl in (1, 2, 0):
        m.set_planar(pl)
        res[pl] = {k: v.clone() for k, v in m.infer_tiles(tiles, osz).items()}
        torch.cuda.synchronize()
    for k in res[1]:
        e12 = bool(torch.equal(res[1][k], res[2][k]))
        if not e12:
            a, b = res[1][k].float(), res[2][k].float()
            print("MISMATCH n=%d hw=%d %s: max abs diff %.3e, %d of %d elements differ" % (n, hw, k, float((a - b).abs().max()), int((a != b).sum()), a.numel()), flush=True)
            ok = False
        assert torch.equal(res[1][k], res[0][k]), "planar 1 vs NHWC differ?!"
print("bitwise planar 2 == planar 1 on all cases:", ok, flush=True)
if "--time" in sys.argv or ok:
    for pl in ((2,) if quick else (1, 2, 1, 2)):
        m.set_planar(pl)
        dt, step, nt = bench.batch_loop(m, dev, 0, 20, 3, None, "nccl")
        _, rows = bench.kernel_table(m, step, nt)
        r = [k for k in rows if k["kernel"].startswith("conv_wino4p") or k["kernel"].startswith("conv_wino4s")]
        print("planar %d: step %.3f ms | %s" % (pl, dt / 20 * 1e3, ", ".join("%s x%d %.3f ms (frac %.3f)" % (k["kernel"], k["launches"], k["ms_per_step"], k["frac"]) for k in r)), flush=True)
