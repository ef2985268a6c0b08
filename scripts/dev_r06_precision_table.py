"""Round 6, VERDICT r5 item 1: distance of every 3x3-convolution algorithm from the REFERENCE's float64 evaluation on the reference-generated fixtures,
side by side -- seeded recipe (calibration logits 4 .. 17), every dense head scaled to 30 / 80, structured tiles, and the reference's default init
(650 .. 2200).  Columns: F(4x4,3x3) (conv_algo 6, the default), F(2x2,3x3) (1), direct implicit GEMM (0), and the reference's own fp32 evaluation.
Reads tests/golden/*.npz only (no oracle, no reference)."""
import json
import os
import sys
from collections import OrderedDict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cerberus_amd.net_desc import create_model  # noqa: E402
from cerberus_amd.run_desc import infer_step  # noqa: E402
from cerberus_amd.synth_tiles import structured_tiles  # noqa: E402
from cerberus_amd.weights import default_model_kwargs, make_state_dict, reference_init_state_dict  # noqa: E402

CROPS, CS = [(0, 0), (96, 96), (192, 192)], 64
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def crops(a):
    return np.stack([a[:, y:y + CS, x:x + CS] for (y, x) in CROPS], axis=1)


def main():
    print("%-14s %-12s %9s | %10s %10s %10s | %10s" % ("fixture", "head", "|logit|", "F(4x4)", "F(2x2)", "direct", "ref fp32"))
    for tag in ("cfg2_all", "struct_all", "logit30_all", "logit80_all", "struct80_all", "refinit_all"):
        g = np.load(os.path.join(GOLD, "net_%s.npz" % tag))
        tasks = [str(t) for t in g["tasks"]]
        kw = default_model_kwargs(tasks)
        fam = str(g["weight_family"])
        if fam == "refinit":
            sd = reference_init_state_dict(kw["decoder_kwargs"], kw["considered_tasks"], generator=torch.Generator().manual_seed(int(g["weight_seed"])))
        else:
            scale = {str(k): np.float32(v) for k, v in zip(g["head_scale_names"], g["head_scale_values"])} if fam == "scaled" else None
            sd = make_state_dict(int(g["weight_seed"]), kw["decoder_kwargs"], kw["considered_tasks"], head_logit_scale=scale)
        n, hw, osz = int(g["n"]), int(g["hw"]), int(g["out_shape"])
        if "tiles_kind" in g and str(g["tiles_kind"]) == "structured":
            tiles = structured_tiles(hw, int(g["tile_seed"]))
        else:
            tiles = np.random.RandomState(int(g["tile_seed"])).randint(0, 256, (n, hw, hw, 3)).astype(np.uint8)
        err = OrderedDict()
        for algo in (6, 1, 0):
            m = create_model(**kw)
            m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
            m.set_conv_algo(algo)
            out = infer_step(torch.from_numpy(tiles), m, osz, tasks)
            for k in out[0]:
                if not k.endswith("INST"):
                    continue
                a = crops(np.stack([out[i][k] for i in range(n)]))
                err.setdefault(k, {})[algo] = float(np.abs(a - g["p64_crops/" + k]).max())
            del m
        for k, e in err.items():
            r32 = float(np.abs(g["out_crops/" + k] - g["p64_crops/" + k]).max())
            print("%-14s %-12s %9.1f | %10.2e %10.2e %10.2e | %10.2e" % (tag, k, float(g["logit_absmax/" + k]), e[6], e[1], e[0], r32))


if __name__ == "__main__":
    main()
