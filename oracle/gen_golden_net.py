"""Generate tests/golden/net_*.npz by running the REFERENCE itself (CPU) in this container.

Run (py3.10 + torch):  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_net.py
Needs /root/reference (read-only); never runs on the GPU box.  The reference's
`NetDesc` / `infer_step` are imported, loaded (strict=True) with the seeded weights of
cerberus_amd.weights.make_state_dict, and run on seeded uint8 tiles.  What is stored is
DATA only: inputs' seeds, crops of the reference's logits, its infer_step outputs
(crops) and whole-tensor float64 statistics.

Inert stubs: cv2 / skimage / termcolor / docopt are not installed here and are only
reached by *imports* of models/run_desc.py -> misc/utils.py, never by infer_step's
arithmetic.  `.to("cuda")` (models/run_desc.py:440) is neutralised
because this container has no GPU.
"""
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

for m in ["cv2", "skimage", "skimage.filters", "skimage.morphology", "termcolor", "matplotlib", "matplotlib.pyplot",
          "tensorboardX", "imgaug", "imgaug.augmenters"]:
    if m not in sys.modules:
        try:
            __import__(m)
        except Exception:
            sys.modules[m] = MagicMock()

_orig_to = torch.Tensor.to


def _to(self, *a, **k):
    if a and a[0] == "cuda":
        return self
    return _orig_to(self, *a, **k)


torch.Tensor.to = _to

from models.net_desc import create_model  # noqa: E402  (reference)
from models.run_desc import infer_step as ref_infer_step  # noqa: E402  (reference)

from cerberus_amd.weights import default_model_kwargs, make_state_dict, reference_init_state_dict, state_dict_sha256  # noqa: E402
from cerberus_amd.synth_tiles import structured_tiles, tiles_sha256  # noqa: E402
from collections import OrderedDict  # noqa: E402
from oracle import net_ref  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)

CROPS = [(0, 0), (96, 96), (192, 192)]  # top-left corners of 64x64 windows (corner / centre / far corner)
CS = 64


def crops(a, axes=(1, 2)):
    """a: (N,H,W,C) -> (N, ncrop, CS, CS, C)"""
    return np.stack([a[:, y:y + CS, x:x + CS] for (y, x) in CROPS], axis=1)


PROBE_SEED = 20240229  # cerberus_amd.net_desc.NetDesc's calibration tile (one seeded 256^2 noise tile)


def calibration_head_scale(kw, weight_seed, target):
    """Per dense head: target / (largest |logit| of the REFERENCE on the calibration tile with the unscaled seeded recipe), as float32 -- the
    factors of make_state_dict(head_logit_scale=...), with which the reference's calibration logits of every dense head land on `target`."""
    sd = {k: torch.from_numpy(v) for k, v in make_state_dict(weight_seed, kw["decoder_kwargs"], kw["considered_tasks"]).items()}
    model = create_model(**kw)
    model.load_state_dict(sd, strict=True)
    model.eval()
    tile = np.random.RandomState(PROBE_SEED).randint(0, 256, (1, 256, 256, 3)).astype(np.uint8)
    with torch.no_grad():
        lg = model(torch.from_numpy(tile).float().permute(0, 3, 1, 2).contiguous())
    out = OrderedDict()
    for dec, heads in kw["decoder_kwargs"].items():
        if dec == "Patch-Class" or dec not in kw["considered_tasks"]:
            continue
        for clf in heads:
            key = dec.split("#")[0] + "-" + clf
            out[dec + "." + clf] = np.float32(target / float(lg[key].abs().max()))
    return out


def run_case(tag, tile_seed, n, hw, out_shape, tasks, weight_seed=0, family="seeded", logit_target=None, tiles_kind="noise", decoder_kwargs=None, head_list=None):
    """family "seeded": cerberus_amd.weights.make_state_dict (non-saturating); "refinit": the distribution the reference's own constructor
    leaves in a fresh model (weights_init_cnn, models/net_desc.py:89-103: kaiming-normal convs, identity BatchNorm) drawn from a seeded
    torch generator (cerberus_amd.weights.reference_init_state_dict) so that the GPU box can rebuild the same tensors."""
    kw = default_model_kwargs(tasks)
    if decoder_kwargs is not None:  # another decoder / head layout than models/paramset.yml's (several heads over one decoder: models/net_desc.py:81-87)
        kw["decoder_kwargs"] = OrderedDict((k, OrderedDict(v)) for k, v in decoder_kwargs)
    head_list = list(kw["considered_tasks"]) if head_list is None else list(head_list)  # infer_step's head_name_list (models/run_desc.py:475-476)
    head_scale = None
    if family == "refinit":
        sd_np = reference_init_state_dict(kw["decoder_kwargs"], kw["considered_tasks"], generator=torch.Generator().manual_seed(weight_seed))
    elif family == "scaled":  # "a confident trained model": the seeded recipe with every dense head's last 1x1 scaled so that its calibration logits reach logit_target
        head_scale = calibration_head_scale(kw, weight_seed, logit_target)
        sd_np = make_state_dict(weight_seed, kw["decoder_kwargs"], kw["considered_tasks"], head_logit_scale=head_scale)
    else:
        sd_np = make_state_dict(weight_seed, kw["decoder_kwargs"], kw["considered_tasks"])
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    model = create_model(**kw)
    model.load_state_dict(sd, strict=True)  # pins the key schema too
    model.eval()
    if tiles_kind == "structured":  # stain field / half glass / all white / all black (cerberus_amd.synth_tiles): what a slide feeds the network besides texture
        tiles = structured_tiles(hw, tile_seed)
        assert tiles.shape[0] == n
    else:
        tiles = np.random.RandomState(tile_seed).randint(0, 256, (n, hw, hw, 3)).astype(np.uint8)
    x = torch.from_numpy(tiles).float().permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        ref_logits = model(x)
        ref_feats = model.backbone(x / 255.0)
    ref_out = ref_infer_step(torch.from_numpy(tiles), model, out_shape, head_list)

    # oracle vs reference (same machine, same torch) -- must agree to rounding
    orc_logits = net_ref.net_forward(sd, x, kw["decoder_kwargs"], kw["considered_tasks"])
    orc_out = net_ref.infer_step(sd, tiles, out_shape, head_list, kw["decoder_kwargs"])
    store = {"tile_seed": tile_seed, "weight_seed": weight_seed, "weight_family": family, "n": n, "hw": hw, "out_shape": out_shape,
             "tasks": np.array(tasks), "weights_sha256": state_dict_sha256(sd_np), "tiles_kind": tiles_kind, "tiles_sha256": tiles_sha256(tiles)}
    if decoder_kwargs is not None:
        import json

        store["decoder_kwargs_json"] = json.dumps([[k, list(v.items())] for k, v in kw["decoder_kwargs"].items()])
        store["head_name_list"] = np.array(head_list)
    if head_scale is not None:
        store["head_scale_names"] = np.array(list(head_scale.keys()))
        store["head_scale_values"] = np.array(list(head_scale.values()), np.float32)
        store["logit_target"] = np.float64(logit_target)
    # The reference's OWN rounding noise: the same model and tiles evaluated in float64 (model.double()), read out like infer_step does
    # (softmax, channels 1..2 of INST heads, argmax of TYPE heads).  noise/<head> = max |p_fp32 - p_fp64| over the whole tensor: two faithful
    # fp32 evaluations of this network cannot be expected to agree more closely than this.  margin/<head>: top-1 minus top-2 softmax
    # probability of the float32 reference per pixel (crops) -- where it is tiny an argmax may legitimately flip.
    import copy

    m64 = copy.deepcopy(model).double()
    with torch.no_grad():
        lg64 = m64(x.double())
    for k, v in ref_logits.items():
        p32 = torch.softmax(v.double(), 1)
        p64 = torch.softmax(lg64[k], 1)
        store["noise/" + k] = np.float64((p32 - p64).abs().max().item())
        store["logit_noise_rel/" + k] = np.float64(((v.double() - lg64[k]).abs().max() / lg64[k].abs().max()).item())
        store["logit_absmax/" + k] = np.float64(lg64[k].abs().max().item())
        top = torch.topk(torch.softmax(v, 1), 2, dim=1).values
        mg = (top[:, 0] - top[:, 1]).numpy()[..., None]  # (N, H, W, 1)
        if k != "Patch-Class" and mg.shape[1] > out_shape:  # the kept window of infer_step (cropping_center)
            o = (mg.shape[1] - out_shape) // 2
            mg = mg[:, o:o + out_shape, o:o + out_shape]
        if k != "Patch-Class":
            store["margin/" + k] = crops(mg) if mg.shape[1] >= 256 else mg
        if k.endswith("INST"):
            # the float64 evaluation itself, read out like infer_step reads the float32 one (softmax channels 1..2, centre crop): the anchor of
            # tests/test_net_gpu.py's bar |got - p64| <= max|ref_fp32 - p64| + 1e-4 (rounded to float32 for storage: 6e-8)
            pr = p64[:, 1:3].permute(0, 2, 3, 1).contiguous().numpy()
            if pr.shape[1] > out_shape:
                o = (pr.shape[1] - out_shape) // 2
                pr = pr[:, o:o + out_shape, o:o + out_shape]
            store[("p64_crops/" if pr.shape[1] >= 256 else "p64_full/") + k] = (crops(pr) if pr.shape[1] >= 256 else pr).astype(np.float32)
        print("%-12s %-12s reference fp32-vs-fp64: probabilities %.3e, logits %.3e relative (|logit| max %.1f)" %
              (tag, k, store["noise/" + k], store["logit_noise_rel/" + k], store["logit_absmax/" + k]))
    for k, v in ref_logits.items():
        d = (orc_logits[k] - v).abs().max().item()
        print("%-12s logits %-18s absmax %.4f  oracle-vs-ref maxdiff %.3e" % (tag, tuple(v.shape), v.abs().max().item(), d))
        assert d < 2e-4 * max(1.0, v.abs().max().item() / 10.0), (k, d)
        a = v.permute(0, 2, 3, 1).contiguous().numpy()
        if a.shape[1] >= 256:
            store["logits_crops/" + k] = crops(a)
        else:
            store["logits_full/" + k] = a
        store["logits_mean/" + k] = np.float64(a.astype(np.float64).mean())
        store["logits_absmean/" + k] = np.float64(np.abs(a.astype(np.float64)).mean())
    for i, f in enumerate(ref_feats):
        a = f.numpy().astype(np.float64)
        store["feat_mean/x%d" % i] = a.mean()
        store["feat_absmean/x%d" % i] = np.abs(a).mean()
    for i in range(n):
        for k, v in ref_out[i].items():
            o = orc_out[i][k]
            assert o.shape == v.shape and o.dtype == v.dtype, (k, o.shape, v.shape, o.dtype, v.dtype)
            if v.dtype == np.float32:
                d = np.abs(o - v).max()
                assert d < max(1e-5, 2.0 * float(store["noise/" + k])), (k, d)
            else:
                mism = (o != v).mean()
                assert mism < 1e-3, (k, mism)
    for k in ref_out[0].keys():
        a = np.stack([ref_out[i][k] for i in range(n)])
        store["out_dtype/" + k] = str(a.dtype)
        if a.ndim == 3:
            a4 = a[..., None]
        else:
            a4 = a
        if a.shape[1] >= 256:
            store["out_crops/" + k] = crops(a4)
        else:
            store["out_full/" + k] = a4
        if a.dtype == np.float32 and "INST" in k:
            sat = np.mean((a < 1e-6) | (a > 1 - 1e-6))
            fg = np.mean(a[..., 0] > 0.5)
            print("%-12s out %-12s saturated frac %.4f  inner>0.5 frac %.3f" % (tag, k, sat, fg))
        if "TYPE" in k:
            print("%-12s out %-12s class hist %s" % (tag, k, np.bincount(a.ravel(), minlength=3)))
        if k == "Patch-Class":
            print("%-12s Patch-Class ids %s" % (tag, a[:, 0, 0]))
    path = os.path.join(ROOT, "tests", "golden", "net_%s.npz" % tag)
    np.savez_compressed(path, **store)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    all_tasks = list(default_model_kwargs()["considered_tasks"])
    only = set(sys.argv[1:])  # optional: regenerate the named fixtures only

    def run_case(tag, _f=run_case, **k):
        if not only or tag in only:
            _f(tag, **k)

    # cfg-1: one 256^2 tile, nuclei head only (BASELINE.json configs[0])
    run_case("cfg1_nuclei", tile_seed=0, n=1, hw=256, out_shape=256, tasks=["Nuclei"])
    # cfg-2 subset: 2 tiles, all six heads
    run_case("cfg2_all", tile_seed=1, n=2, hw=256, out_shape=256, tasks=all_tasks)
    # reference-default geometry 448 -> 144 (centre crop path of infer_step)
    run_case("g448_all", tile_seed=2, n=1, hw=448, out_shape=144, tasks=all_tasks)
    # bottom feature map smaller than 9 x 9 (6 x 6): cropping_center's negative-start slice in the Patch-Class branch
    run_case("small96_all", tile_seed=3, n=2, hw=96, out_shape=96, tasks=all_tasks)
    # a second draw of the same recipe
    run_case("seed1_all", tile_seed=4, n=2, hw=256, out_shape=256, tasks=all_tasks, weight_seed=1)
    # the reference's default initialisation: logits in the hundreds / thousands, saturated probabilities -- every fp32 evaluation is far
    # from the fp64 one here (noise/<head> in the fixture); the regime DESIGN.md par.4.0 calls the stress case
    run_case("refinit_all", tile_seed=5, n=2, hw=256, out_shape=256, tasks=all_tasks, weight_seed=0, family="refinit")
    # round 6: the band between "seeded" (calibration logits 4 .. 17) and "refinit" (650 .. 2200) -- a confident trained
    # model: every dense head's calibration logits at 30 and at 80 -- and structured inputs (stain field, half glass, white, black) on the
    # seeded weights and on the logit-80 family
    run_case("logit30_all", tile_seed=6, n=2, hw=256, out_shape=256, tasks=all_tasks, weight_seed=0, family="scaled", logit_target=30.0)
    run_case("logit80_all", tile_seed=7, n=2, hw=256, out_shape=256, tasks=all_tasks, weight_seed=0, family="scaled", logit_target=80.0)
    run_case("struct_all", tile_seed=8, n=4, hw=256, out_shape=256, tasks=all_tasks, weight_seed=0, tiles_kind="structured")
    # several output heads over ONE decoder (models/net_desc.py:81-87, 196-198): the Gland decoder carries INST and TYPE
    run_case("multihead", tile_seed=10, n=2, hw=256, out_shape=256, tasks=["Gland", "Nuclei", "Patch-Class"],
             decoder_kwargs=[("Gland", [("INST", 3), ("TYPE", 3)]), ("Nuclei", [("INST", 3)]), ("Patch-Class", [("OUT", 9)])],
             head_list=["Gland", "Gland#TYPE", "Nuclei", "Patch-Class"])
    run_case("struct80_all", tile_seed=9, n=4, hw=256, out_shape=256, tasks=all_tasks, weight_seed=0, family="scaled", logit_target=80.0, tiles_kind="structured")
