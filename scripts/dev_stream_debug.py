"""Developer: why does the streamed labelling differ?  Compare run_local on the resident canvases cut at the sub-band rows with infer_and_label_streamed."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_drivers_gpu import _PixelNet  # noqa: E402

from cerberus_amd import synth_maps  # noqa: E402
from cerberus_amd.shard_postproc import band_view, run_local  # noqa: E402
from cerberus_amd.stream_bands import infer_and_label_streamed  # noqa: E402
from cerberus_amd.wsi import WSIRunner  # noqa: E402

H, W, margin = 1500, 1300, 256
nuc, gl = synth_maps.nuclei_maps(H, W, 5, 1500.0), synth_maps.blob_maps(H, W, 6, 40, 24.0, 50.0, rim=4.0, sharp=1.0)
slide = torch.from_numpy(np.stack([nuc[..., 0], nuc[..., 1], gl[..., 0]], -1).clip(0, 1) * 255.0).to(torch.uint8).cuda()
net = _PixelNet()
run = WSIRunner(net, (H, W), 256, 256, batch_size=4)
run.infer_band(slide, 0)
bv = band_view(run, H, W)
full = bv["Nuclei-INST"]
print("canvas equals the map:", float((full[..., 0] - slide[..., 0].float() / 255).abs().max()))
outs, n, infos = run_local([full], "Nuclei", margin, 16)
print("one band:", n, infos)
cuts = [0, 512, 1024, 1500]
outs3, n3, infos3 = run_local([full[cuts[i]:cuts[i + 1]] for i in range(3)], "Nuclei", margin, 16)
print("three bands (run_local):", n3, infos3)
got, info, small = infer_and_label_streamed(net, lambda a, b: slide[a:b].contiguous(), (H, W), 256, 256, 4, 3, margin=margin, guard=16)
print("streamed:", info["Nuclei"])
print("equal to one band:", torch.equal(got["Nuclei"], outs[0]), "run_local3 equal:", torch.equal(torch.cat(outs3), outs[0]))
