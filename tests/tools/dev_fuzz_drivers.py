"""Developer fuzz: tile manager / WSI runner / sharding / sharded post-processing on random image sizes and patch geometries."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tests/: test_drivers_gpu._oracle_stitch
import numpy as np, torch
from cerberus_amd import shard_postproc as sp
from cerberus_amd.postproc import postproc_device
from cerberus_amd.tile import InferManager
from cerberus_amd.weights import DEFAULT_REQ_TARGET_CODE, default_model_kwargs, make_state_dict
from cerberus_amd.wsi import WSIRunner, synth_slide
from oracle import synth
from test_drivers_gpu import _oracle_stitch

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
kw = default_model_kwargs()
mgr = InferManager(checkpoint_path=None, decoder_dict=dict(DEFAULT_REQ_TARGET_CODE), model_args=kw)
sd = {k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}
bad = 0
for i in range(n_cases):
    win = 16 * int(rs.randint(9, 20))
    out = int(rs.randint(win // 3, win + 1)); out -= (win - out) % 2
    H, W = int(rs.randint(40, 700)), int(rs.randint(40, 700))
    slide = synth_slide(H, W, seed=int(rs.randint(100)))
    img = slide.cpu().numpy()
    res = mgr.infer_image(img, win, out, batch_size=int(rs.randint(1, 9)))
    msg = []
    if i % 2 == 0:  # oracle stitch is the slow part: every other case
        ref = _oracle_stitch(img, win, out, kw, sd)
        for k, r in ref.items():
            a = res["raw"][k].cpu().numpy()
            if a.shape != r.shape: msg.append("shape " + k)
            elif r.dtype == np.float32 and np.abs(a - r).max() > 1e-4: msg.append("%s err %.1e" % (k, np.abs(a - r).max()))
            elif r.dtype != np.float32 and (a != r).mean() > 1e-4: msg.append("%s mismatch %.1e" % (k, (a != r).mean()))
    world = int(rs.randint(1, 5))
    one = WSIRunner(mgr.net, (H, W), win, out, batch_size=5); one.infer_band(slide, 0); full = one.gather_to_root()
    # tile mode pads like numpy 1.x (iterative, see tile._pad_reflect_numpy1); the resident-slide gather mirrors periodically:
    # the two agree whenever no pad exceeds the image side minus one
    from cerberus_amd.tile import _prepare_patching as _pp
    _padded, _, _pos = _pp(img, win, out, 0)
    single_reflection = max(_padded.shape[0] - H - _pos[0], _pos[0]) <= H - 1 and max(_padded.shape[1] - W - _pos[1], _pos[1]) <= W - 1
    for k, v in full.items():
        if single_reflection and not torch.equal(v, res["raw"][k]): msg.append("wsi!=tile " + k)
    parts = {}
    ok_world = True
    for r in range(world):
        run = WSIRunner(mgr.net, (H, W), win, out, batch_size=4, rank=r, world_size=world)
        if run.r1 <= run.r0: ok_world = False; break
        y0, y1 = run.slab_rows()
        run.infer_band(slide[y0:y1].contiguous(), y0)
        for k, v in run.canv.items(): parts.setdefault(k, []).append(v.clone())
    if ok_world:
        for k, v in full.items():
            if not torch.equal(torch.cat(parts[k], 0)[:H, :W], v): msg.append("shard!=whole " + k)
    print("case %2d: img %3dx%3d win %3d out %3d world %d %s" % (i, H, W, win, out, world, "FAIL " + "; ".join(msg) if msg else "ok"), flush=True)
    bad += bool(msg)
# sharded post-processing: random band splits of structured maps
for i in range(n_cases):
    tissue = ["Nuclei", "Gland", "Lumen"][i % 3]
    ds = 1.0 if tissue == "Nuclei" else float(rs.choice([1.0, 0.5]))
    H, W = int(rs.randint(900, 1600)), int(rs.randint(300, 900))
    seed = int(rs.randint(1 << 20))
    m = synth.nuclei_maps(H, W, seed, float(rs.choice([300, 900])), noise=0.02) if tissue == "Nuclei" else \
        synth.blob_maps(H, W, seed, max(4, H * W // 20000), 8.0, 30.0, rim=3.0, sharp=1.0, noise=0.02, holes=0.3)
    full = torch.from_numpy(m).cuda()
    ref, info = postproc_device(full, tissue, ds)
    world = int(rs.randint(2, 5))
    margin = 160
    cuts = sorted(rs.choice(np.arange(margin + 8, H - margin - 8), world - 1, replace=False).tolist())
    bounds = [0] + cuts + [H]
    if min(b - a for a, b in zip(bounds[:-1], bounds[1:])) < margin:
        print("pp case %d: skipped (band shorter than margin)" % i); continue
    bands = [full[a:b] for a, b in zip(bounds[:-1], bounds[1:])]
    outs, n_total, infos = sp.run_local(bands, tissue, margin, 24, ds)
    lab = sp.assemble(outs).cpu().numpy()
    trunc = sum(x["n_truncated"] + x["n_unresolved"] for x in infos)
    same = sp.same_partition(ref.cpu().numpy(), lab)
    status = "ok" if same else ("differs, %d truncated/unresolved flagged" % trunc if trunc else "FAIL (unflagged difference)")
    bad += (not same and trunc == 0)
    print("pp case %2d: %s %dx%d ds %.1f bounds %s n %d %s" % (i, tissue, H, W, ds, bounds, n_total, status), flush=True)
print("driver fuzz: %d failures" % bad)
