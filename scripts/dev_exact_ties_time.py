"""How often the tie rule fires on saturated-core maps and what the heap replay costs (GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cerberus_amd.postproc import postproc_device
from cerberus_amd import synth_maps as synth

def timed(m, exact):
    postproc_device(m, "Nuclei", exact_ties=exact); torch.cuda.synchronize()
    t = time.perf_counter()
    lab, info = postproc_device(m, "Nuclei", exact_ties=exact); torch.cuda.synchronize()
    return lab, int(info["n_ambiguous"].item()), (time.perf_counter() - t) * 1e3

for S in (1024, 2048, 4096):
    for dens in (600.0, 1500.0, 3000.0):
        for gain in (4.0, 8.0, 20.0):
            m = torch.from_numpy(synth.softmax_nuclei_maps(S, S, 7, dens, gain=gain, logit_noise=0.5)).cuda()
            sat = float((m[..., 0] == 1.0).float().mean())
            fast, amb, tf = timed(m, False)
            if amb:
                ex, _, te = timed(m, True)
                nd = int((ex != fast).sum())
                npx = int((ex > 0).sum())
            else:
                te, nd, npx = float("nan"), 0, int((fast > 0).sum())
            print("%5d^2 dens %5.0f gain %4.1f: saturated %.3f | n_inst %6d mask px %8d | flagged regions %4d (%.1f / Mpx) | fast %.2f ms | with replay %.1f ms (%.2f us / mask px), %d px change"
                  % (S, dens, gain, sat, int(fast.max()), npx, amb, amb / (S * S / 1e6), tf, te, (te - tf) * 1e3 / max(npx, 1), nd), flush=True)
