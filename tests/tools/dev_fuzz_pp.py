"""Developer fuzz: random map sizes / densities / noise levels / ds factors, HIP post-processing vs the C oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cerberus_amd.postproc import postproc_device
from oracle import postproc_ref as pr, synth

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1234)
bad = 0
amb_cases = 0
t_start = time.time()
for i in range(n_cases):
    tissue = ["Nuclei", "Nuclei", "Gland", "Lumen"][rs.randint(4)]
    H, W = int(rs.randint(17, 900)), int(rs.randint(17, 900))
    seed = int(rs.randint(1 << 30))
    noise = float(rs.choice([0.0, 0.02, 0.1, 0.3]))
    if tissue == "Nuclei":
        dens = float(rs.choice([100, 600, 2000, 6000]))
        kind = rs.randint(4)
        if kind == 0:
            m = synth.nuclei_maps(H, W, seed, dens, noise=noise)
        elif kind == 1:  # saturated plateaus: heavy ties
            m = np.round(synth.nuclei_maps(H, W, seed, dens, noise=noise) * 4) / 4
        elif kind == 2:  # large touching blobs
            m = synth.blob_maps(H, W, seed, max(3, H * W // 6000), 6.0, 30.0, rim=2.0, sharp=float(rs.choice([0.5, 1.5])), noise=noise, border_bias=True)
        else:  # pure noise field
            m = rs.rand(H, W, 2).astype(np.float32) * np.array([1.2, 0.4], np.float32)
        ds = 1.0
    else:
        ds = float(rs.choice([1.0, 0.5, 0.3])) if tissue == "Gland" else float(rs.choice([1.0, 0.5]))
        m = synth.blob_maps(H, W, seed, max(2, H * W // int(rs.choice([3000, 10000, 40000]))), 5.0, float(rs.choice([15, 40, 90])), rim=float(rs.choice([2.0, 4.0])),
                            sharp=1.0, noise=noise, holes=float(rs.choice([0.0, 0.3, 0.7])), border_bias=bool(rs.randint(2)))
    m = np.ascontiguousarray(m.astype(np.float32))
    ref = pr.proc(m, tissue, ds).astype(np.int64)
    got, info = postproc_device(torch.from_numpy(m).cuda(), tissue, ds)
    got = got.cpu().numpy().astype(np.int64)
    amb = int(info["n_ambiguous"].item()) if tissue == "Nuclei" else 0
    mism = int((got != ref).sum())
    if mism and amb == 0:
        bad += 1
        print("MISMATCH case %d: %s %dx%d seed %d noise %.2f ds %.1f -> %d px differ, n_ref %d n_got %d" % (i, tissue, H, W, seed, noise, ds, mism, ref.max(), got.max()), flush=True)
    elif mism:
        amb_cases += 1
print("fuzz: %d cases, %d mismatching with n_ambiguous == 0, %d cases differ only where ties are flagged; %.1f s" % (n_cases, bad, amb_cases, time.time() - t_start))
