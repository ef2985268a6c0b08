"""Developer fuzz (CPU only): the flood's ambiguity RULES (tests/tools/tie_rule_sim.c, the same rules the HIP kernels apply)
against the oracle's literal skimage heap on tie-heavy maps.  Contract: ambiguous == 0  =>  identical label map."""
import ctypes as C
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np

from oracle import postproc_ref as pr, synth

so = os.path.join(HERE, "_tie_rule_sim.so")
src = os.path.join(HERE, "tie_rule_sim.c")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, src, "-lm"])
L = C.CDLL(so)
L.sim_proc_nuclei.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]


def sim(m, mode):
    H, W = m.shape[:2]
    out = np.zeros((H, W), np.int32)
    amb = C.c_int(0)
    L.sim_proc_nuclei(m.ctypes.data, H, W, out.ctypes.data, mode, C.byref(amb))
    return out, amb.value


def make(rs, kind, H, W):
    seed = int(rs.randint(1 << 30))
    noise = float(rs.choice([0.0, 0.02, 0.1, 0.3]))
    dens = float(rs.choice([100, 600, 2000, 6000]))
    if kind == 0:
        return synth.nuclei_maps(H, W, seed, dens, noise=noise)
    if kind == 1:
        q = int(rs.choice([1, 2, 3, 4, 8, 16, 64]))
        return np.round(synth.nuclei_maps(H, W, seed, dens, noise=noise, sharp=float(rs.choice([0.6, 1.5, 6.0]))) * q) / q
    if kind == 2:
        return synth.blob_maps(H, W, seed, max(3, H * W // 6000), 6.0, 30.0, rim=2.0, sharp=float(rs.choice([0.5, 1.5])), noise=noise, border_bias=True)
    if kind == 3:
        return rs.rand(H, W, 2).astype(np.float32) * np.array([1.2, 0.4], np.float32)
    if kind == 4:
        return synth.softmax_nuclei_maps(H, W, seed, dens, gain=float(rs.choice([2.0, 4.0, 8.0, 20.0])), logit_noise=float(rs.choice([0.0, 0.5, 2.0])))
    if kind == 5:
        return synth.nuclei_maps(H, W, seed, dens, noise=noise).astype(np.float16).astype(np.float32)
    # noise field quantised: ties everywhere, holes, small components next to markers
    q = int(rs.choice([2, 4, 16]))
    return np.round(rs.rand(H, W, 2).astype(np.float32) * np.array([1.2, 0.4], np.float32) * q) / q


n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
maxhw = int(sys.argv[3]) if len(sys.argv) > 3 else 300
stats = {}
t0 = time.time()
for i in range(n_cases):
    kind = int(rs.randint(7))
    H, W = int(rs.randint(9, maxhw)), int(rs.randint(9, maxhw))
    m = np.ascontiguousarray(make(rs, kind, H, W).astype(np.float32))
    ref = pr.proc(m, "Nuclei")
    s = stats.setdefault(kind, dict(n=0, old_amb=0, new_amb=0, old_mis=0, new_mis=0, bad_old=0, bad_new=0))
    s["n"] += 1
    for mode, tag in ((0, "old"), (1, "new")):
        got, amb = sim(m, mode)
        mis = int((got != ref).sum())
        s[tag + "_amb"] += amb > 0
        s[tag + "_mis"] += mis > 0
        if mis and amb == 0:
            s["bad_" + tag] += 1
            print("VIOLATION rule %s: case %d kind %d %dx%d -> %d px differ with ambiguous == 0" % (tag, i, kind, H, W, mis), flush=True)
print("kind: cases | flagged old/new | maps differing from skimage order old/new | violations old/new   (%.0f s)" % (time.time() - t0))
for k in sorted(stats):
    s = stats[k]
    print("%d: %4d | %4d %4d | %4d %4d | %d %d" % (k, s["n"], s["old_amb"], s["new_amb"], s["old_mis"], s["new_mis"], s["bad_old"], s["bad_new"]))
