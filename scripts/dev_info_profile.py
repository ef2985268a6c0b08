"""Developer probe: where the time of the instance dictionary goes on a structured nuclei map (table / contours / host dict / joblib)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cerberus_amd.postproc import postproc_device, inst_table_device, inst_contours_device, get_inst_info_dict
from cerberus_amd.wsi import build_wsi_inst_info
from cerberus_amd import synth_maps as synth
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = torch.from_numpy(synth.nuclei_maps(H, W, 7, 1000.0, noise=0.02)).cuda()
tm = torch.randint(0, 7, (H, W), dtype=torch.uint8, device="cuda")
lab, info = postproc_device(m, "Nuclei")
n = int(info["n_inst"])
def T(f, rep=2):
    for _ in range(rep):
        torch.cuda.synchronize(); t0 = time.time(); r = f(); torch.cuda.synchronize(); dt = time.time() - t0
    return r, dt
tab, dt = T(lambda: inst_table_device(lab, tm, n)); print("inst_table %.1f ms" % (dt * 1e3), flush=True)
_, dt = T(lambda: inst_contours_device(lab, tab)); print("contours %.1f ms" % (dt * 1e3), flush=True)
d, dt = T(lambda: get_inst_info_dict(lab, tm)); print("get_inst_info_dict %.1f ms for %d instances (%.2f us each)" % (dt * 1e3, len(d), dt * 1e6 / max(len(d), 1)), flush=True)
w, dt = T(lambda: build_wsi_inst_info({"Nuclei": lab}, {"Nuclei-TYPE": tm}, (H, W), 0.5), 1); print("build_wsi_inst_info %.1f ms" % (dt * 1e3), flush=True)
import joblib
from cerberus_amd.wsi import write_dat
p = os.path.join(tempfile.mkdtemp(), "x.dat")
t0 = time.time(); write_dat(w, p); print("write_dat %.1f ms, %.1f MB" % ((time.time() - t0) * 1e3, os.path.getsize(p) / 1e6), flush=True)
t0 = time.time(); b = joblib.load(p); print("joblib.load %.1f ms, %d nuclei" % ((time.time() - t0) * 1e3, len(b["Nuclei"])), flush=True)
