"""Developer probe: wall time of one gradient step (train-mode forward + losses + backward) and of a whole train_step at growing sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cerberus_amd.losses import PARAMSET_LOSS
from cerberus_amd.net_desc import create_model
from cerberus_amd.weights import default_model_kwargs, make_state_dict
m = create_model(**default_model_kwargs())
m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}, strict=True)
heads = {"Lumen-INST": 3, "Gland-INST": 3, "Nuclei-INST": 3, "Nuclei-TYPE": 7, "Gland-TYPE": 3, "Patch-Class": 9}
for n, hw in [(2, 64), (4, 128), (4, 256), (16, 448)][: int(sys.argv[1]) if len(sys.argv) > 1 else 3]:
    rs = np.random.RandomState(0)
    tiles = torch.from_numpy(rs.randint(0, 256, (n, hw, hw, 3)).astype(np.uint8)).cuda()
    tg = {h: torch.from_numpy((rs.randint(0, c, (n,)) if h == "Patch-Class" else (rs.rand(n, hw, hw) < 0.3) * rs.randint(1, c, (n, hw, hw))).astype(np.float32)).cuda()
          for h, c in heads.items()}
    fl = {h: torch.ones(n).cuda() for h in heads}
    for rep in range(2):  # the first call allocates the tape and the workspaces: report the second
        torch.cuda.synchronize(); t0 = time.time()
        losses, grads = m.train_grads(tiles, tg, fl, PARAMSET_LOSS, None)
        torch.cuda.synchronize(); t1 = time.time() - t0
        t0 = time.time(); m.forward_train(tiles); torch.cuda.synchronize(); t2 = time.time() - t0
    print("batch %2d x %3d^2: gradient step %.3f s (train-mode forward alone %.3f s), overall loss %.4f, peak memory %.1f GB" % (
        n, hw, t1, t2, sum(losses.values()), torch.cuda.max_memory_allocated() / 1e9), flush=True)
    from cerberus_amd.train import Adam, train_step
    if not hasattr(m, "_opt"): m._opt = Adam(lr=1.0e-4, betas=(0.9, 0.999))
    batch = {"img": tiles.cpu(), "dummy_target": np.array([list(heads)] * n, dtype=object)}
    for h in heads: batch[h] = tg[h].cpu().reshape((n,) if h == "Patch-Class" else (n, hw, hw, 1))
    for rep in range(3):  # whole step as the reference's engine calls it: host batch in, losses out, Adam + BN statistics + weight re-pack
        torch.cuda.synchronize(); t0 = time.time()
        res = train_step(batch, ({"net": {"desc": m, "optimizer": m._opt, "extra_info": {"loss": PARAMSET_LOSS}}}, None))
        torch.cuda.synchronize(); t3 = time.time() - t0
    print("    whole train_step (host batch in, Adam, running statistics, weight re-pack): %.3f s, overall loss %.4f" % (t3, res["EMA"]["overall_loss"]), flush=True)
