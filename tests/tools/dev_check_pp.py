import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cerberus_amd.postproc import postproc_device
from oracle import postproc_ref as pr, synth
g = np.load("tests/golden/pp_cases.npz")
for name in [str(n) for n in g["names"]]:
    m = g["in/" + name].astype(np.float32); tissue = str(g["tissue/" + name]); ds = float(g["ds/" + name]); ref = g["out/" + name]
    got, info = postproc_device(torch.from_numpy(m).cuda(), tissue, ds)
    got = got.cpu().numpy()
    print("%-18s %-7s mismatches %6d  n_inst %4d (ref max %4d)  ambiguous %d" % (name, tissue, (got != ref).sum(), int(info["n_inst"]), ref.max(), int(info["n_ambiguous"])))
for hw, dens in [((1024, 1024), 1500.0), ((2048, 2048), 1000.0), ((4096, 4096), 600.0)]:
    m = synth.nuclei_maps(hw[0], hw[1], 7, dens, noise=0.02)
    t0 = time.time(); ref = pr.proc(m, "Nuclei"); tc = time.time() - t0
    md = torch.from_numpy(m).cuda()
    postproc_device(md, "Nuclei"); torch.cuda.synchronize()
    t0 = time.time(); got, info = postproc_device(md, "Nuclei"); torch.cuda.synchronize(); tg = time.time() - t0
    print("nuclei %s: inst %d  mismatches %d  amb %d  cpu %.3fs (%.1f Mpx/s)  gpu %.4fs (%.1f Mpx/s)" % (hw, ref.max(), (got.cpu().numpy() != ref).sum(), int(info["n_ambiguous"]), tc, hw[0]*hw[1]/tc/1e6, tg, hw[0]*hw[1]/tg/1e6))
for tissue in ("Gland", "Lumen"):
    m = synth.blob_maps(2048, 2048, 9, 120, 14.0, 60.0, rim=4.0, sharp=1.0, noise=0.02, holes=0.3)
    t0 = time.time(); ref = pr.proc(m, tissue); tc = time.time() - t0
    md = torch.from_numpy(m).cuda()
    postproc_device(md, tissue); torch.cuda.synchronize()
    t0 = time.time(); got, info = postproc_device(md, tissue); torch.cuda.synchronize(); tg = time.time() - t0
    print("%s 2048^2: inst %d mismatches %d cpu %.3fs gpu %.4fs (%.1f Mpx/s)" % (tissue, ref.max(), (got.cpu().numpy() != ref.astype(np.int32)).sum(), tc, tg, 4.19/tg))
